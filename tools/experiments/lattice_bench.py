"""Dense-lattice window kernel (u3d_igemm_lattice_bf16) against the neighbour-table kernel on the bench workload's dense layers:
parity (same operands) and timing.  usage (GPU): python tools/lattice_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni3detr_amd import native as nv  # noqa: E402

CASES = [("dense256 k27", 8, (15, 40, 40), 256, 256, (3, 3, 3)), ("dense256 k9 ", 8, (15, 20, 20), 256, 256, (1, 3, 3)),
         ("dense512 k9 ", 8, (15, 10, 10), 512, 512, (1, 3, 3))]


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
    for name, B, dims, cin, cout, ks in CASES:
        n = B * dims[0] * dims[1] * dims[2]
        kvol = ks[0] * 9
        pad = (ks[0] // 2, 1, 1)
        nbr = nv.dense_nbr_table(B, dims, dims, ks, (1, 1, 1), pad, 0, dev)
        nbr_b = nv.dense_nbr_table(B, dims, dims, ks, (1, 1, 1), pad, 1, dev)
        nd = nv.count_tensor(n, dev)
        torch.manual_seed(0)
        x = torch.randn(n, cin, device=dev).bfloat16()
        dy = torch.randn(n, cout, device=dev).bfloat16()
        kio = (torch.randn(kvol, cin, cout, device=dev) * 0.05).bfloat16()
        koi = kio.transpose(1, 2).contiguous()
        flops = 2.0 * float((nbr[:, :n] >= 0).sum()) * cin * cout
        ref_f = nv.spconv_fwd(x, koi, nbr, nd, n, cout, transpose_w=True, tag="spconv_fwd")
        ref_d = nv.spconv_fwd(dy, kio, nbr_b, nd, n, cin, transpose_w=True)
        got_f = nv.lattice_conv(x, koi, B, dims, ks[0])
        got_d = nv.lattice_conv(dy, kio, B, dims, ks[0], transposed=True)
        if got_f is None:
            print(name, "not served")
            continue
        ef = (got_f.float() - ref_f.float()).abs().max().item() / ref_f.float().abs().max().item()
        ed = (got_d.float() - ref_d.float()).abs().max().item() / ref_d.float().abs().max().item()
        _, st = nv.lattice_conv(x, koi, B, dims, ks[0], want_stats=True)
        es = (st[:, 0].sum(0).float() - got_f.float().sum(0)).abs().max().item()
        t_tab = timeit(lambda: nv.spconv_fwd(x, koi, nbr, nd, n, cout, transpose_w=True, tag="spconv_fwd"))
        t_lat = timeit(lambda: nv.lattice_conv(x, koi, B, dims, ks[0]))
        t_tabd = timeit(lambda: nv.spconv_fwd(dy, kio, nbr_b, nd, n, cin, transpose_w=True))
        t_latd = timeit(lambda: nv.lattice_conv(dy, kio, B, dims, ks[0], transposed=True))
        print(f"{name} N={n}: fwd table {t_tab * 1e3:7.1f} us ({flops / t_tab / 1e9:6.0f} TF/s)  lattice {t_lat * 1e3:7.1f} us ({flops / t_lat / 1e9:6.0f} TF/s) | "
              f"dgrad table {t_tabd * 1e3:7.1f}  lattice {t_latd * 1e3:7.1f} | rel diff fwd {ef:.1e} dgrad {ed:.1e} stats {es:.2e}", flush=True)


if __name__ == "__main__":
    main()
