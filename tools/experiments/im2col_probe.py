"""Would the strided dense convs (SECOND3D first conv of each branch) run faster as im2col + a plain GEMM?  Times the GEMM / weight
gradient halves on the existing kernels with contiguous rows of K * Cin elements (the im2col gather itself is priced as a copy)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni3detr_amd import native as nv  # noqa: E402


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for n_out, cin, cout in ((12000, 256, 512), (48000, 256, 256)):
    kc = 9 * cin
    a = torch.randn(n_out, kc, device="cuda").bfloat16()
    w = (torch.randn(1, cout, kc, device="cuda") * 0.02).bfloat16()
    dy = torch.randn(n_out, cout, device="cuda").bfloat16()
    nd = nv.count_tensor(n_out, "cuda")
    src = torch.randn(192000, cin, device="cuda").bfloat16()
    idx = torch.randint(0, 192000, (n_out * 9,), device="cuda")
    print(f"n_out {n_out} {cin}->{cout}: fwd+stats {t(lambda: nv.spconv_fwd_stats(a, w, None, nd, n_out, cout)):7.1f} us | "
          f"wgrad {t(lambda: nv.spconv_wgrad(a, dy, None, nd, 1)):7.1f} us | gather (index_select stand-in) {t(lambda: src.index_select(0, idx)):7.1f} us")
