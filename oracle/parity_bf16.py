"""bf16-mode deviation of the product training forward from the fp32 CPU oracle (oracle/model.py) on the benched workload shape.

Checker code (test infrastructure, like everything under oracle/): used by tests/test_bf16_parity_gpu.py and by bench.py's
cpu_baseline leg, never by the product path.
The reference keeps the sparse encoder and the dense backbone in fp32 and runs neck + head under fp16 autocast (ref:
models/pts_encoder/sparse_encoder_hd.py:63-64, models/detectors/uni3detr.py:150-151); the throughput mode here runs all of them in
bf16 with f32 accumulation, so what is measured is the total effect of bf16 storage on the quantities training consumes.
"""
import numpy as np
import torch


def rel_l2(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-12))


def _scenes(B, npts):
    from uni3detr_amd.synth import room_scene
    scenes = [room_scene(i, npts - 2500 * i) for i in range(B)]
    pts = [torch.from_numpy(s[0]) for s in scenes]
    gtb = []
    for s in scenes:
        g = torch.from_numpy(s[1]).clone()
        g[:, 2] -= g[:, 5] / 2
        gtb.append(g)
    return pts, gtb, [torch.from_numpy(s[2]) for s in scenes]


def oracle_reference(B=2, npts=20000, seed=11):
    """The fp32 CPU oracle's training forward on the deviation workload: (losses, aux) - computed once, shared by every mode."""
    import projects.mmdet3d_plugin  # noqa: F401
    from oracle import model as om
    from oracle.weights import seeded_tensor
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    sd = {k: seeded_tensor(k, tuple(v.shape), seed) for k, v in build_model(MODEL_CFG).state_dict().items()}
    pts, gtb, labels = _scenes(B, npts)
    with torch.no_grad():
        ref_losses, aux = om.forward_train(sd, [p.numpy() for p in pts], gtb, labels, om.sunrgbd_cfg())
    return dict(losses=ref_losses, aux=aux, B=B, npts=npts, seed=seed)


def bf16_deviation(device, B=2, npts=20000, seed=11, mode="bf16", ref=None):
    """mode: "bf16" (throughput mode), "mixed" (the reference's recipe: fp32 encoder + backbone, 16-bit neck + head), "parity"
    (f32 storage everywhere, convolutions as split-bf16 products, exact-f32 decoder) or "fp32" (exact-f32 MFMA everywhere).
    Returns dict: feature / logit relative L2 deviations, share of identical Hungarian assignments, worst relative loss deviation.
    ref: oracle_reference(...) of the same (B, npts, seed) - computed here when absent."""
    import projects.mmdet3d_plugin  # noqa: F401
    from oracle import model as om
    from oracle.weights import seeded_tensor
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.plugin.structures import Boxes3D
    from uni3detr_amd.registry import build_model

    if ref is None:
        ref = oracle_reference(B, npts, seed)
    assert (ref["B"], ref["npts"], ref["seed"]) == (B, npts, seed)
    ref_losses, aux = ref["losses"], ref["aux"]
    model = build_model(MODEL_CFG)
    sd = {k: seeded_tensor(k, tuple(v.shape), seed) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "attn_drop"):
            m.attn_drop = 0.0
    model = model.to(device).train()
    model.set_precision(mode)
    pts, gtb, labels = _scenes(B, npts)
    extra = {}
    if mode in ("mixed", "parity"):
        # the modules the reference keeps in fp32 (encoder + backbone), on their own: f32 rows, wide convs as split-bf16 products
        with torch.no_grad():
            from uni3detr_amd import sparse as sp
            v = model.stage_voxelize([p.to(device) for p in pts])
            with sp.split_scope(True):
                enc = model.pts_middle_encoder(v["feats"], v["fcoors"], len(v["lens"]))
                bb = model.pts_backbone(enc)
            extra = {"encoder_rel_l2": rel_l2(enc, aux["encoder"]), "backbone_rel_l2": max(rel_l2(a, b) for a, b in zip(bb, aux["backbone"])),
                     "encoder_dtype": str(enc.dtype)}
    with model.shadow_scope():
        feat, fpsb = model.extract_pts_feat([p.to(device) for p in pts])
        amp = model.amp_dtype
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            outs = model.pts_bbox_head(feat, None, fpsb)
        losses = model.pts_bbox_head.loss([Boxes3D(g).to(device) for g in gtb], [l.to(device) for l in labels], outs)
    asg = model.pts_bbox_head._last_assigned.cpu()
    out = {
        "fps_queries_identical": bool(torch.equal(fpsb.cpu(), aux["fpsbpts"])),
        "feature_rel_l2": rel_l2(feat, aux["feats"]),
        "cls_logit_rel_l2": rel_l2(outs["all_cls_scores"], aux["cls"]),
        "box_rel_l2": rel_l2(outs["all_bbox_preds"], aux["box"]),
        "iou_logit_rel_l2": rel_l2(outs["all_iou_preds"], aux["iou"]),
        "cls_logit_max_abs": float((outs["all_cls_scores"].detach().float().cpu() - aux["cls"]).abs().max()),
        "assignments_identical_share": float((asg == aux["assigned"]).float().mean()),
        "matched_assignments_identical_share": float((asg == aux["assigned"])[aux["assigned"] > 0].float().mean()),
        "loss_max_rel": max(abs(float(losses[k].detach()) - float(v)) / max(1.0, abs(float(v))) for k, v in ref_losses.items()),
        "loss_total_rel": abs(float(sum(losses.values()).detach()) - float(sum(ref_losses.values()))) / abs(float(sum(ref_losses.values()))),
        "workload": f"{B} scenes x {npts} pts, seeded weights, dropout off", "mode": mode,
    }
    out.update(extra)
    return out
