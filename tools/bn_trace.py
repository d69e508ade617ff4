"""BatchNorm row kernels on ONE tensor shape, 30 iterations, no timing of its own: run it under
`rocprofv3 --kernel-trace --stats -- python tools/bn_trace.py <rows> <channels>` to get DEVICE durations per kernel
(tools/bn_bench.py's event timings are host-bound below ~50 MB: the ctypes call + two allocations take longer than the kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni3detr_amd import native as nv
n, c = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
x = torch.randn(n, c, device=dev).bfloat16(); dy = torch.randn(n, c, device=dev).bfloat16()
nd = nv.count_tensor(n, dev)
gamma = torch.rand(c, device=dev) + 0.5; beta = torch.randn(c, device=dev) * 0.1
sums = nv.bn_stats(x, nd); mean, invstd = nv.bn_finalize(sums, nd, n, 1e-3, 0.1)
for _ in range(30):
    bs = nv.bn_bwd_stats(dy, None, x, mean, invstd, True, nd, gamma, beta)
    nv.bn_bwd_apply(dy, None, x, mean, invstd, gamma, bs, True, nd, False, beta)
    nv.bn_apply(x, mean, invstd, gamma, beta, None, True, nd)
torch.cuda.synchronize()
