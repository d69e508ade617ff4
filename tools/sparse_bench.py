"""Per-level timing of the sparse-encoder convolutions on REAL neighbour tables (GPU): the bench workload's scenes are voxelized,
the encoder's levels are built (uni3detr_amd/sparse.py), and every (level, pass) the step runs is timed in isolation.
usage: python tools/sparse_bench.py [--iters 30] [--only 32] [--check]
tools/conv_bench.py prices the same kernels on full lattices (every neighbour present); here ~1/3 of the pairs exist."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from uni3detr_amd import native as nv  # noqa: E402
from uni3detr_amd import sparse as sp  # noqa: E402


def ref_fwd(x, w, nbr, n):
    out = torch.zeros(n, w.shape[2], dtype=torch.float32, device=x.device)
    xf = torch.cat([x.float(), torch.zeros(1, x.shape[1], device=x.device)])
    for k in range(w.shape[0]):
        idx = nbr[k, :n].long()
        idx = torch.where(idx < 0, torch.full_like(idx, x.shape[0]), idx)
        out += xf[idx] @ w[k].float()
    return out


def build_jobs(batch, dev):
    """[(name, cin, cout, ConvGeom)] of the bench workload's sparse levels."""
    import projects.mmdet3d_plugin  # noqa: F401
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    model = build_model(MODEL_CFG).to(dev).train()
    data = bench.make_batch(0, batch, 20000, dev)
    coors = model.voxelize_batch(data["points"])[0]
    enc = model.pts_middle_encoder
    lvl, _ = sp.level_from_coors(coors.int().contiguous(), batch, enc.sparse_shape)
    # (name, cin, cout, geom): the SubM convs of a level share one table; the strided conv into the next level has its own
    jobs = []
    cin = enc.base_channels
    for i, blocks in enumerate(enc.encoder_channels):       # block_type 'basicblock': SubM blocks, then the strided conv into the next level
        blocks = tuple(blocks)
        jobs.append((f"L{i} subm N={lvl.n}", cin, cin, sp.subm_geom(lvl)))
        if i != len(enc.encoder_channels) - 1:
            pad = tuple(enc.encoder_paddings[i])[len(blocks) - 1]
            pad = (pad,) * 3 if isinstance(pad, int) else tuple(pad)
            st = enc.encoder_strides[i]
            st = (st,) * 3 if isinstance(st, int) else tuple(st)
            new, geom = sp.strided_level(lvl, (3, 3, 3), st, pad)
            jobs.append((f"L{i} down N={lvl.n}->{new.n}", cin, blocks[-1], geom))
            lvl = new
            cin = blocks[-1]
    return jobs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default=None)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    jobs = build_jobs(a.batch, dev)
    torch.manual_seed(0)
    for name, ci, co, g in jobs:
        tagname = f"{name} {ci}->{co}"
        if a.only and a.only not in tagname:
            continue
        x = torch.relu(torch.randn(g.n_in, ci, device=dev)).bfloat16()
        dy = torch.randn(g.n_out, co, device=dev).bfloat16()
        w = (torch.randn(27, ci, co, device=dev) * 0.05).bfloat16()
        pairs = int((g.nbr_fwd[:, :g.n_out] >= 0).sum())
        flops = 2.0 * pairs * ci * co
        byt = g.n_in * ci * 2 + g.n_out * co * 2 + 8 * pairs + 27 * ci * co * 2
        passes = {
            "fwd": lambda: nv.spconv_fwd(x, w, g.nbr_fwd, g.n_out_dev, g.n_out, co),
            "dgrad": lambda: nv.spconv_fwd(dy, w, g.nbr_bwd, g.n_in_dev, g.n_in, ci, transpose_w=True),
            "wgrad": lambda: nv.spconv_wgrad(x, dy, g.nbr_fwd, g.n_out_dev, 27),
        }
        if os.environ.get("HALO_STATS") and g.level is not None and ci == co:
            h_ = nv.SubmHalo(g.nbr_fwd, g.n_out_dev, g.n_out)
            tc_ = h_.tile_cnt.float()
            print(f"{tagname:34s} halo stats: {h_.tiles} tiles, distinct rows per tile mean {tc_.mean().item():.0f} p90 {tc_.quantile(0.9).item():.0f} max {tc_.max().item():.0f}", flush=True)
        if ci == 64 and co == 64 and g.level is not None:
            halo = g.level.halo()
            tc = halo.tile_cnt.float()
            print(f"{tagname:34s} halo: {halo.tiles} tiles, distinct rows per tile mean {tc.mean().item():.0f} p50 {tc.median().item():.0f} "
                  f"p90 {tc.quantile(0.9).item():.0f} max {tc.max().item():.0f}", flush=True)
            wn = w.transpose(1, 2).contiguous()
            passes["fwd_nmajor_stats"] = lambda: nv.spconv_fwd_stats(x, wn, g.nbr_fwd, g.n_out_dev, g.n_out, co)
            passes["halo_build"] = lambda: nv.SubmHalo(g.nbr_fwd, g.n_out_dev, g.n_out)
            wpf, wpb = nv.subm_halo_wpack(wn), nv.subm_halo_wpack(w)
            passes["halo_wpack"] = lambda: nv.subm_halo_wpack(wn)
            passes["halo_fwd"] = lambda: nv.subm_halo_conv(x, wpf, halo)
            passes["halo_fwd_stats"] = lambda: nv.subm_halo_conv(x, wpf, halo, want_stats=True)
            passes["halo_dgrad"] = lambda: nv.subm_halo_conv(dy, wpb, halo, krev=True)
            passes["halo_wgrad"] = lambda: nv.subm_halo_wgrad(x, dy, halo)
        if ci == co and ci in (32, 128) and g.level is not None and g.level.halo() is not None:
            halo = g.level.halo()
            wn = w.transpose(1, 2).contiguous()
            wpf, wpb = nv.subm_halo_wpack(wn), nv.subm_halo_wpack(w)
            passes["fwd_nmajor_stats"] = lambda: nv.spconv_fwd_stats(x, wn, g.nbr_fwd, g.n_out_dev, g.n_out, co)
            passes["halo_fwd"] = lambda: nv.subm_halo_conv(x, wpf, halo)
            passes["halo_fwd_stats"] = lambda: nv.subm_halo_conv(x, wpf, halo, want_stats=True)
            passes["halo_dgrad"] = lambda: nv.subm_halo_conv(dy, wpb, halo, krev=True)
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
            print(f"{tagname:34s} {pname:6s} pairs/row {pairs / max(1, g.n_out):5.1f}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s  {byt / ms / 1e6:7.0f} GB/s (algorithmic)", flush=True)
        if a.check:
            m = min(g.n_out, 8192)
            got = nv.spconv_fwd(x, w, g.nbr_fwd, g.n_out_dev, g.n_out, co)[:m].float()
            exp = ref_fwd(x, w, g.nbr_fwd, m)
            err = (got - exp).abs().max().item() / max(1.0, exp.abs().max().item())
            mi = min(g.n_in, 8192)
            gd = nv.spconv_fwd(dy, w, g.nbr_bwd, g.n_in_dev, g.n_in, ci, transpose_w=True)[:mi].float()
            nb = g.nbr_bwd.t.flip(0) if isinstance(g.nbr_bwd, nv.RevNbr) else g.nbr_bwd
            ed = ref_fwd(dy, w.transpose(1, 2), nb, mi)
            errd = (gd - ed).abs().max().item() / max(1.0, ed.abs().max().item())
            gw = nv.spconv_wgrad(x, dy, g.nbr_fwd, g.n_out_dev, 27)
            errw = 0.0
            xf = torch.cat([x.float(), torch.zeros(1, ci, device=dev)])
            for k in (0, 13, 26):
                idx = g.nbr_fwd[k, :g.n_out].long()
                idx = torch.where(idx < 0, torch.full_like(idx, g.n_in), idx)
                ew = xf[idx].t() @ dy.float()
                errw = max(errw, (gw[k] - ew).abs().max().item() / max(1.0, ew.abs().max().item()))
            print(f"{tagname:34s} check: fwd {err:.2e} dgrad {errd:.2e} wgrad {errw:.2e}", flush=True)
            assert err < 2e-2 and errd < 2e-2 and errw < 2e-2


if __name__ == "__main__":
    main()
