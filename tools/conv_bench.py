"""Per-layer timing of the implicit-GEMM kernels on the bench workload's layer shapes (GPU).
usage: python tools/conv_bench.py [--only dense256] [--iters 20] [--check]
Prints one line per (layer, pass): ms, TFLOP/s (pair-based flops), algorithmic GB/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni3detr_amd import native as nv  # noqa: E402

LAYERS = {
    # name: (batch, dims(z,y,x), cin, cout, ksize, stride, pad)
    "dense256": (8, (15, 40, 40), 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "dense128k9": (8, (15, 40, 40), 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "dense256k9": (8, (15, 20, 20), 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "dense512k9": (8, (15, 10, 10), 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "dense64": (8, (15, 40, 40), 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "dense32": (8, (30, 60, 24), 32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "dense32to64": (8, (30, 60, 24), 32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "dense16to32": (8, (20, 40, 20), 16, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "dense16": (8, (20, 40, 20), 16, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # the same 128-channel layer at other heights (KITTI: 704 000 rows): how the two-phase tile's efficiency depends on the launch size
    "rows96k_128k9": (4, (15, 40, 40), 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "rows384k_128k9": (16, (15, 40, 40), 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "rows704k_128k9": (4, (5, 200, 176), 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    # SECOND3D's strided first convs (branches 1 and 2 read the 192 000-row volume with stride 2 / 4)
    "stride2_256": (8, (15, 40, 40), 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    "stride4_512": (8, (15, 40, 40), 256, 512, (1, 3, 3), (1, 4, 4), (0, 1, 1)),
}


def ref_fwd(x, w, nbr, n):
    out = torch.zeros(n, w.shape[2], dtype=torch.float32, device=x.device)
    xf = torch.cat([x.float(), torch.zeros(1, x.shape[1], device=x.device)])
    for k in range(w.shape[0]):
        idx = nbr[k, :n].long()
        idx = torch.where(idx < 0, torch.full_like(idx, x.shape[0]), idx)
        out += xf[idx] @ w[k].float()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--data", default="randn", choices=["randn", "relu", "small", "zeros", "bn"], help="value distribution of the activation / gradient operands (kernel time depends on operand values through power/clock management)")
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many input/gradient tensors so the working set exceeds the 256 MB Infinity Cache (the training step never re-reads a tensor it just used)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, (B, dims, cin, cout, ks, st, pad) in LAYERS.items():
        if a.only and a.only not in name:
            continue
        dims_out = tuple((d + 2 * p_ - k_) // s_ + 1 for d, k_, s_, p_ in zip(dims, ks, st, pad))
        n_in = B * dims[0] * dims[1] * dims[2]
        n = B * dims_out[0] * dims_out[1] * dims_out[2]
        kvol = ks[0] * ks[1] * ks[2]
        nbr = nv.dense_nbr_table(B, dims_out, dims, ks, st, pad, 0, dev)
        nbr_b = nv.dense_nbr_table(B, dims, dims_out, ks, st, pad, 1, dev)
        nd = nv.count_tensor(n, dev)
        nd_in = nv.count_tensor(n_in, dev)
        torch.manual_seed(0)
        def gen(c, rows=None):
            t = torch.randn(n if rows is None else rows, c, device=dev)
            if a.data == "relu":
                t = torch.relu(t)
            elif a.data == "small":
                t = t * 1e-3
            elif a.data == "zeros":
                t = t * 0
            elif a.data == "bn":
                t = torch.relu(t * 0.7 + 0.1)
            return t.bfloat16()
        xs = [gen(cin, n_in) for _ in range(a.rotate)]
        dys = [gen(cout) for _ in range(a.rotate)]
        x, dy = xs[0], dys[0]
        ctr = [0]

        def nxt():
            ctr[0] += 1
            return xs[ctr[0] % a.rotate], dys[ctr[0] % a.rotate]
        w = (torch.randn(kvol, cin, cout, device=dev) * 0.05).bfloat16()
        pairs = int((nbr[:, :n] >= 0).sum())
        flops = 2.0 * pairs * cin * cout
        byt = n * cin * 2 + n * cout * 2 + 8 * pairs + kvol * cin * cout * 2
        koi = w.transpose(1, 2).contiguous()             # n-major [K][Cout][Cin]: what the training step's forward passes (sparse._SparseConv)
        passes = {
            "fwd": lambda: nv.spconv_fwd(nxt()[0], koi, nbr, nd, n, cout, transpose_w=True, tag="spconv_fwd"),
            "fwd_k": lambda: nv.spconv_fwd(nxt()[0], w, nbr, nd, n, cout),         # k-major weights: the inference / first-generation path
            "dgrad": lambda: nv.spconv_fwd(nxt()[1], w, nbr_b, nd_in, n_in, cin, transpose_w=True),
            "wgrad": lambda: nv.spconv_wgrad(*nxt(), nbr, nd, kvol),
        }
        for pname, fn in passes.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"{name:12s} {pname:6s} N={n} {cin}->{cout} K={kvol}: {ms:8.4f} ms  {flops / ms / 1e9:8.1f} TF/s  {byt / ms / 1e6:8.1f} GB/s", flush=True)
        if a.check:
            m = min(n, 4096)
            got = nv.spconv_fwd(x, w, nbr, nd, n, cout)[:m].float()
            exp = ref_fwd(x, w, nbr, m)
            err = (got - exp).abs().max().item() / max(1.0, exp.abs().max().item())
            gd = nv.spconv_fwd(dy, w, nbr_b, nd, n, cin, transpose_w=True)[:m].float()
            ed = ref_fwd(dy, w.transpose(1, 2), nbr_b, m)
            errd = (gd - ed).abs().max().item() / max(1.0, ed.abs().max().item())
            gw = nv.spconv_wgrad(x, dy, nbr, nd, kvol)
            errw = 0.0
            xf = torch.cat([x.float(), torch.zeros(1, cin, device=dev)])
            for k in sorted({0, kvol // 2, kvol - 1}):
                idx = nbr[k, :n].long()
                idx = torch.where(idx < 0, torch.full_like(idx, n), idx)
                ew = xf[idx].t() @ dy.float()
                errw = max(errw, (gw[k] - ew).abs().max().item() / max(1.0, ew.abs().max().item()))
            print(f"{name:12s} check: fwd rel err {err:.2e}  dgrad rel err {errd:.2e}  wgrad rel err {errw:.2e}", flush=True)
            assert err < 2e-2 and errd < 2e-2 and errw < 2e-2


if __name__ == "__main__":
    main()
