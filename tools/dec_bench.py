"""Fused decoder layer in isolation: `iters` forward + backward calls of one layer at the bench shape (8 scenes x 900 queries, dropout
on), for kernel-level timing under a rocprofv3 kernel trace (tools/dec_bench.sh) and for A/B runs of kernel variants
(U3D_LIB_PATH=uni3detr_amd/_variants/<name>.so).  Prints wall time per forward+backward pair as a sanity figure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import projects.mmdet3d_plugin  # noqa: F401
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.plugin import fused_decoder as fdm
from uni3detr_amd.registry import build_model


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    et = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    head = build_model(MODEL_CFG).pts_bbox_head.to(dev).train()
    eval_mode = os.environ.get("U3D_DEC_EVAL") == "1"          # no dropout, forward only: the attention kernel without its hash / mask VALU work
    if eval_mode:
        head = head.eval()
    dec = head.transformer.decoder
    fd = fdm.FusedDecoder(dec, head.reg_branches, head.cls_branches, head.iou_branches)
    B, G, nq, D, H, W = 8, 3, 300, 15, 40, 40
    M = B * G * nq
    lid = 1
    sp = fd.specs[lid]
    x = (torch.randn(M, 256, device=dev) * 0.7).requires_grad_(True)
    ref = torch.randn(M, 3, device=dev) * 1.2
    rows = torch.randn(B * D * H * W, 256, device=dev).clamp_min(0).to(et).requires_grad_(True)
    fd.refresh(dev, et)
    meta = (fd, lid, (B, G * nq, nq, D, H, W), None, et)
    plist = fdm.tensor_list(sp)

    def once():
        if eval_mode:
            with torch.no_grad():
                fdm.FusedLayerFn.apply(x, None, ref, rows, meta, *plist)
            return
        outs = fdm.FusedLayerFn.apply(x, None, ref, rows, meta, *plist)
        loss = outs[0].sum() + outs[2].sum() + outs[3].sum() + outs[4].sum()
        loss.backward()
        x.grad = None
        rows.grad = None
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    print(f"dec_bench {et}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per fwd+bwd (host wall, includes torch glue)")
    # -DDC_PHASE_TIMING builds: shader-clock stamps of one workgroup (wave 0) at the phase boundaries of the two post kernels
    import ctypes as C
    from uni3detr_amd import native as nv
    lib = C.CDLL(nv.LIB_PATH)
    FWD = ['start', 'outp gemm', 'LN1', 'gate+gather', 'pe0 valu', 'oproj gemm + LN pe0', 'pe1 gemm', 'LN pe1 + add', 'LN2', 'ffn0 gemm x2',
           'ffn1 gemm', 'LN3', 'reg branch + iou branch', 'cls branch']
    BWD = ['start', 'cls branch (narrow dgrad, 2 LN bwd, 2 gemm)', 'iou branch (narrow dgrad, 2 gemm)', 'reg branch (narrow dgrad, 2 gemm)',
           'LN3 bwd', 'ffn: branch_grad + 2 gemm (N=512) + gemm K=512', 'LN2 bwd', 'pos encoder: LN bwd x2 + gemm', 'oproj gemm',
           'gate bwd + scatter atomics', 'LN1 bwd + store + outp gemm']
    for sym, names in (("u3d_debug_fwd_times", FWD), ("u3d_debug_bwd_times", BWD)):
        if not hasattr(lib, sym):
            continue
        buf = (C.c_uint64 * 64)()
        getattr(lib, sym)(buf)
        t = list(buf)[:len(names) + 1]
        tot = t[-1] - t[0]
        print(f"-- {sym}: {tot} shader clocks in the marked workgroup")
        full = list(buf)
        if sym == "u3d_debug_fwd_times" and full[41] > full[32] > 0:
            mp = full[32:42]
            names_mp = ["issue row loads + weight copy", "barrier (weights landed)", "in-projection: 3 row blocks", "barrier (images complete)",
                        "slot stores (q, k, v)", "query block 0", "query block 1", "query block 2", "tail"]
            print(f"-- k_mha_proj_fwd, workgroup 7 wave 0: {mp[9] - mp[0]} shader clocks")
            for i, n in enumerate(names_mp):
                if mp[i + 1] >= mp[i] > 0:
                    print(f"   {n:50s} {mp[i + 1] - mp[i]:8d} clk")
        if sym == "u3d_debug_fwd_times" and full[23] > full[20] > 0:
            print(f"   [reg0 linear taken apart] burst+MFMA {full[21] - full[20]}  epilogue {full[22] - full[21]}  barrier {full[23] - full[22]} clk")
        for i, n in enumerate(names):
            print(f"   {n:50s} {t[i + 1] - t[i]:8d} clk  {100.0 * (t[i + 1] - t[i]) / tot:5.1f} %")


if __name__ == "__main__":
    main()
