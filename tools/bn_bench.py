"""Row-op (BatchNorm) kernel timings on the bench workload's tensor shapes (GPU): ms and achieved GB/s per kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni3detr_amd import native as nv  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    for n, c in [(192000, 256), (192000, 128), (48000, 256), (12000, 512), (338532, 32), (207393, 64), (128000, 16)]:
        torch.manual_seed(0)
        x = torch.randn(n, c, device=dev).bfloat16()
        dy = torch.randn(n, c, device=dev).bfloat16()
        res = torch.randn(n, c, device=dev).bfloat16()
        nd = nv.count_tensor(n, dev)
        gamma = torch.rand(c, device=dev) + 0.5
        beta = torch.randn(c, device=dev) * 0.1
        sums = nv.bn_stats(x, nd)
        mean, invstd = nv.bn_finalize(sums, nd, n, 1e-3, 0.1)
        y = nv.bn_apply(x, mean, invstd, gamma, beta, None, True, nd)
        bs = nv.bn_bwd_stats(dy, None, x, mean, invstd, True, nd, gamma, beta)
        S = n * c * 2
        cases = [
            ("stats", lambda: nv.bn_stats(x, nd), 1),
            ("apply", lambda: nv.bn_apply(x, mean, invstd, gamma, beta, None, True, nd), 2),
            ("apply+res", lambda: nv.bn_apply(x, mean, invstd, gamma, beta, res, True, nd), 3),
            ("bwd_stats(remask)", lambda: nv.bn_bwd_stats(dy, None, x, mean, invstd, True, nd, gamma, beta), 2),
            ("bwd_stats(y)", lambda: nv.bn_bwd_stats(dy, y, x, mean, invstd, True, nd, gamma, beta), 3),
            ("bwd_apply(remask)", lambda: nv.bn_bwd_apply(dy, None, x, mean, invstd, gamma, bs, True, nd, False, beta), 3),
            ("bwd_apply(y,dres)", lambda: nv.bn_bwd_apply(dy, y, x, mean, invstd, gamma, bs, True, nd, True, beta), 5),
        ]
        for name, fn, k in cases:
            ms = timeit(fn)
            print(f"N={n:7d} C={c:4d} {name:20s} {ms * 1e3:8.1f} us  {k * S / ms / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
