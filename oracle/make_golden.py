"""ORACLE (test infrastructure).  Generates tests/golden/*.npz by running the REFERENCE's own files
(/root/reference/projects/mmdet3d_plugin/..., loaded in place through oracle/refshim.py) on seeded inputs with
name-seeded weights (oracle/weights.py).  Run in the build container only:  python -m oracle.make_golden
Only inputs/outputs (data) are stored; no reference source travels.
"""
import os
import sys

import numpy as np
import torch

from . import refshim as rs
from .weights import seeded_input, seeded_state_dict

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEED = 7


def build_ref_head(ns):
    m = rs.sunrgbd_head_cfg()
    hc = dict(m["pts_bbox_head"])
    hc.pop("type")
    hc["train_cfg"] = m["train_cfg"]["pts"]
    head = ns.head.Uni3DETRHead(**hc)
    sd = seeded_state_dict(head.state_dict().items(), SEED)
    head.load_state_dict(sd)
    head.eval()    # dropout off; query layout switches on requires_grad, not on .training
    return head, {"pts_bbox_head." + k: v for k, v in sd.items()}


def scene_gts(B, n_boxes=8):
    from uni3detr_amd.synth import room_scene
    gts, labels = [], []
    for b in range(B):
        _, gt, l = room_scene(b, 2000, n_boxes=n_boxes - b)      # ragged GT counts
        g = torch.from_numpy(gt).clone()
        g[:, 2] -= g[:, 5] / 2                                     # `.tensor` carries bottom-centre z
        gts.append(g)
        labels.append(torch.from_numpy(l))
    return gts, labels


def head_inputs(B):
    feats = seeded_input("pts_feats", (B, 256, 15, 40, 40), SEED, -0.5, 1.0).clamp_min(0)
    fps = seeded_input("fpsbpts", (B, 600, 3), SEED, 0.0, 1.0)
    return feats, fps


def gen_head_train(ns):
    B = 2
    head, _ = build_ref_head(ns)
    feats, fps = head_inputs(B)
    feats.requires_grad_(True)
    outs = head(feats, None, fps)
    gts, labels = scene_gts(B)
    losses = head.loss([rs.GTBoxes(g) for g in gts], labels, outs)
    total = sum(losses.values())
    total.backward()
    # assignment per layer/scene, re-derived through the reference assigner
    gc = [torch.cat([g[:, :2], g[:, 2:3] + g[:, 5:6] * 0.5, g[:, 3:]], 1) for g in gts]
    assigned = np.zeros((3, B, 900), np.int16)
    costs0 = None
    for l in range(3):
        for b in range(B):
            r = head.assigner.assign(outs["all_bbox_preds"][l, b], outs["all_cls_scores"][l, b], gc[b], labels[b], 300, None, 1)
            assigned[l, b] = r.gt_inds.numpy()
    pgrad = {k: p.grad for k, p in head.named_parameters() if p.grad is not None}
    names = sorted(pgrad)
    np.savez_compressed(
        os.path.join(OUT, "head_train_b2.npz"), seed=SEED,
        cls=outs["all_cls_scores"].detach().numpy(), box=outs["all_bbox_preds"].detach().numpy(),
        iou=outs["all_iou_preds"].detach().numpy(), assigned=assigned,
        loss_names=np.array(sorted(losses)), loss_values=np.array([float(losses[k]) for k in sorted(losses)], np.float64),
        feats_grad_sub=feats.grad.reshape(-1)[::997].numpy(), feats_grad_abs_sum=float(feats.grad.abs().sum()),
        pgrad_names=np.array(names), pgrad_l2=np.array([float(pgrad[k].norm()) for k in names], np.float64),
        sd_names=np.array(list(head.state_dict().keys())),
        sd_shapes=np.array([",".join(str(int(d)) for d in v.shape) for v in head.state_dict().values()]),
        gt_lens=np.array([g.shape[0] for g in gts]), gts=torch.cat(gts).numpy(), labels=torch.cat(labels).numpy())
    print("head_train_b2: losses", {k: round(float(v), 5) for k, v in losses.items()})


def gen_head_eval(ns):
    """Eval layout: 4 groups incl. a random-point group (uni3detr_head.py:446-449); the random points are part of the
    fixture because they come from torch's global RNG."""
    B = 1
    head, _ = build_ref_head(ns)
    feats, fps = head_inputs(B)
    torch.manual_seed(123)
    rand_state = torch.random.get_rng_state()
    rand_pts = torch.rand(fps.shape)[:, :300, :]
    torch.random.set_rng_state(rand_state)
    with torch.no_grad():
        outs = head(feats, None, fps)
    np.savez_compressed(os.path.join(OUT, "head_eval_b1.npz"), seed=SEED, rand_points=rand_pts.numpy(),
                        cls=outs["all_cls_scores"].numpy(), box=outs["all_bbox_preds"].numpy(), iou=outs["all_iou_preds"].numpy())
    print("head_eval_b1:", outs["all_cls_scores"].shape)


def gen_small(ns):
    rng = np.random.default_rng(SEED)
    # box codes
    boxes = np.concatenate([rng.uniform(-3, 3, (64, 3)), rng.uniform(0.2, 2.5, (64, 3)), rng.uniform(-4, 4, (64, 1))], 1).astype(np.float32)
    tb = torch.from_numpy(boxes)
    norm = ns.util.normalize_bbox(tb, None)
    den = ns.util.denormalize_bbox(norm, None)
    # match costs through the reference classes
    pred = torch.from_numpy(np.concatenate([rng.uniform(-3, 3, (50, 2)), rng.uniform(-1, 1, (50, 2)), rng.uniform(-2, 0.5, (50, 1)),
                                            rng.uniform(-1, 1, (50, 3))], 1).astype(np.float32))
    l1 = ns.match_cost.BBox3DL1Cost(0.25)(pred, norm[:12, :8])
    iouc = ns.match_cost.IoU3DCost(1.2)(ns.util.denormalize_bbox(pred, None), tb[:12])
    # sine embedding + MLP-free pieces
    pos = torch.from_numpy(rng.random((2, 37, 3)).astype(np.float32))
    sine = ns.transformer.get_sine_pos_embed(pos)
    # soft focal / iou3d loss
    logits = torch.from_numpy(rng.normal(0, 2, (40, 10)).astype(np.float32))
    lab = torch.from_numpy(rng.integers(0, 11, 40))
    score = torch.from_numpy(rng.random(40).astype(np.float32))
    sfl = ns.losses.soft_focal_loss(logits, [lab, score], torch.ones(40), 2.0, 0.25, "mean", 7.0)
    il = ns.losses.iou3d_loss(tb[:32], tb[32:], None, reduction="none")
    np.savez_compressed(os.path.join(OUT, "small_ops.npz"), boxes=boxes, norm=norm.numpy(), denorm=den.numpy(), pred=pred.numpy(),
                        l1cost=l1.numpy(), ioucost=iouc.numpy(), pos=pos.numpy(), sine=sine.numpy(), logits=logits.numpy(),
                        lab=lab.numpy(), score=score.numpy(), sfl=float(sfl), iou3d_loss=il.numpy())
    print("small_ops ok")


def gen_head_variants(ns):
    """The reference head / transformer with the two other head shapes the shipped configs use: KITTI (9 decoder layers, 3 classes;
    uni3detr_kitti_3classes.py:64-77) through forward + loss + backward, and nuScenes (900 queries -> 2700 training queries, code size
    10; uni3detr_nuscenes.py:69) through the forward only - its loss path is inconsistent upstream (9-dim GT into 7-dim targets)."""
    out = {}
    for name, B, vol, with_loss in (("kitti_3classes", 2, (5, 24, 22), True), ("nuscenes", 1, (5, 20, 20), False)):
        m = rs.model_cfg(name)
        hc = dict(m["pts_bbox_head"])
        hc.pop("type")
        hc["train_cfg"] = m["train_cfg"]["pts"]
        head = ns.head.Uni3DETRHead(**hc)
        sd = seeded_state_dict(head.state_dict().items(), SEED)
        head.load_state_dict(sd)
        head.eval()
        nq = hc["num_query"]
        feats = seeded_input(name + ".pts_feats", (B, 256) + vol, SEED, -0.5, 1.0).clamp_min(0)
        fps = seeded_input(name + ".fpsbpts", (B, 2 * nq, 3), SEED, 0.0, 1.0)
        feats.requires_grad_(True)
        outs = head(feats, None, fps)
        out[name + "_cls"] = outs["all_cls_scores"].detach().numpy()
        out[name + "_box"] = outs["all_bbox_preds"].detach().numpy()
        out[name + "_iou"] = outs["all_iou_preds"].detach().numpy()
        out[name + "_shape"] = np.array((B, 256) + vol + (nq,))
        if with_loss:
            pr = m["pts_bbox_head"]["bbox_coder"]["pc_range"]
            rng = np.random.default_rng(SEED + 5)
            gts, labels = [], []
            for b in range(B):
                n = 6 - b
                c = rng.uniform([pr[0] + 5, pr[1] + 5, pr[2] + 0.5], [pr[3] - 5, pr[4] - 5, pr[5] - 1.5], (n, 3))
                g = np.concatenate([c, rng.uniform(0.6, 4.0, (n, 3)), rng.uniform(-3.1, 3.1, (n, 1))], 1).astype(np.float32)
                gts.append(torch.from_numpy(g))
                labels.append(torch.from_numpy(rng.integers(0, hc["num_classes"], n)))
            losses = head.loss([rs.GTBoxes(g) for g in gts], labels, outs)
            sum(losses.values()).backward()
            out[name + "_loss_names"] = np.array(sorted(losses))
            out[name + "_loss_values"] = np.array([float(losses[k]) for k in sorted(losses)], np.float64)
            out[name + "_gt_lens"] = np.array([g.shape[0] for g in gts])
            out[name + "_gts"] = torch.cat(gts).numpy()
            out[name + "_labels"] = torch.cat(labels).numpy()
            out[name + "_feats_grad_abs_sum"] = float(feats.grad.abs().sum())
            pg = {k: p.grad for k, p in head.named_parameters() if p.grad is not None}
            names = sorted(pg)
            out[name + "_pgrad_names"] = np.array(names)
            out[name + "_pgrad_l2"] = np.array([float(pg[k].norm()) for k in names], np.float64)
        print(name, "head:", out[name + "_cls"].shape, out[name + "_box"].shape)
    np.savez_compressed(os.path.join(OUT, "head_variants.npz"), seed=SEED, **out)


def gen_decode(ns):
    """NMSFreeCoder.decode of the reference file (core/bbox/coders/nms_free_coder.py:42-136) on seeded head outputs: three coder
    settings (alpha, score threshold, a tight centre range) so that the top-k, the score / IoU blend and both masks are pinned."""
    rng = np.random.default_rng(SEED + 1)
    L, B, Q, C = 3, 2, 1200, 10
    cls = torch.from_numpy(rng.normal(-2.0, 2.0, (L, B, Q, C)).astype(np.float32))
    box = torch.from_numpy(np.concatenate([rng.uniform(-3.4, 3.4, (L, B, Q, 1)), rng.uniform(-0.4, 6.4, (L, B, Q, 1)),
                                           rng.normal(0, 0.5, (L, B, Q, 2)), rng.uniform(-2.1, 0.7, (L, B, Q, 1)),
                                           rng.normal(0, 0.5, (L, B, Q, 1)), rng.normal(0, 1, (L, B, Q, 2))], -1).astype(np.float32))
    iou = torch.from_numpy(rng.normal(0, 1.5, (L, B, Q, 1)).astype(np.float32))
    pc = [-3.2, -0.2, -2.0, 3.2, 6.2, 0.56]
    out = dict(cls=cls.numpy(), box=box.numpy(), iou=iou.numpy(), pc_range=np.array(pc, np.float32))
    settings = [dict(alpha=1.0, score_threshold=None, max_num=1000, post_center_range=pc),
                dict(alpha=0.5, score_threshold=0.3, max_num=300, post_center_range=pc),
                dict(alpha=0.25, score_threshold=None, max_num=500, post_center_range=[-2.0, 0.5, -1.5, 2.0, 5.0, 0.3])]
    for si, st in enumerate(settings):
        coder = ns.coder.NMSFreeCoder(pc_range=pc, voxel_size=[0.02, 0.02, 0.02], num_classes=C, **st)
        res = coder.decode(dict(all_cls_scores=cls, all_bbox_preds=box, all_iou_preds=iou))
        out[f"s{si}_cfg"] = np.array([st["alpha"], -1.0 if st["score_threshold"] is None else st["score_threshold"], st["max_num"]] +
                                     list(st["post_center_range"]), np.float64)
        for b, r in enumerate(res):
            for k in ("bboxes", "scores", "labels", "ious"):
                out[f"s{si}_b{b}_{k}"] = r[k].numpy()
    np.savez_compressed(os.path.join(OUT, "coder_decode.npz"), **out)
    print("coder_decode:", [int(out[f"s{si}_b0_scores"].shape[0]) for si in range(len(settings))])


def gen_extra(ns):
    """The registry names no shipped config selects, where the reference implements the arithmetic itself: get_rdiou
    (core/bbox/util.py:104-153), RDIoUCost / SoftFocalLossCost (match_cost.py:69-128), RDIoULoss (rdiouloss.py:13-91)."""
    rng = np.random.default_rng(SEED + 9)
    Q, G, C = 40, 7, 10

    def boxes(n):
        return torch.from_numpy(np.concatenate([rng.uniform(-3, 3, (n, 3)), rng.normal(0, 0.6, (n, 3)), rng.uniform(-3.1, 3.1, (n, 1))], 1)
                                .astype(np.float32))
    p, g = boxes(Q), boxes(G)
    p[0, 3] = 3.0                                             # exp(3) > 10: the clamp of the first box's extents
    u, r = ns.util.get_rdiou(p.unsqueeze(1), g.unsqueeze(0))
    cost = ns.match_cost.RDIoUCost(weight=1.7)(p, g)
    cls = torch.from_numpy(rng.normal(-1, 2, (Q, C)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, C, G))
    iou = torch.from_numpy(rng.uniform(0, 1, (Q, C)).astype(np.float32))
    sfc = ns.match_cost.SoftFocalLossCost(weight=2.0)(cls, labels, iou)
    a, b = boxes(Q), boxes(Q)
    w = torch.from_numpy((rng.uniform(0, 1, (Q, 7)) < 0.6).astype(np.float32))
    a.requires_grad_(True)
    loss = ns.losses.RDIoULoss(loss_weight=1.2)(a, b, w, avg_factor=5.0)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "extra_costs.npz"), p=p.numpy(), g=g.numpy(), u=u.numpy(), rdiou=r.numpy(), rdiou_cost=cost.numpy(),
                        cls=cls.numpy(), labels=labels.numpy(), iou=iou.numpy(), soft_focal_cost=sfc.numpy(), a=a.detach().numpy(), b=b.numpy(),
                        w=w.numpy(), rdiou_loss=float(loss), rdiou_loss_grad=a.grad.numpy())
    print("extra_costs:", tuple(cost.shape), float(loss))


def main():
    if not rs.available():
        sys.exit("reference tree not available: goldens can only be generated in the build container")
    os.makedirs(OUT, exist_ok=True)
    ns = rs.load_hot_path()
    gen_small(ns)
    gen_head_train(ns)
    gen_head_eval(ns)
    gen_decode(ns)
    gen_head_variants(ns)
    gen_extra(ns)


if __name__ == "__main__":
    main()
