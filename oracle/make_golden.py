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


# --------------------------------------------------------------------------------------------------
# round 6: the rest of the hot path that the reference itself implements (VERDICT r5 "What's missing" #2)
# --------------------------------------------------------------------------------------------------
DENSE_CFGS = ("sunrgbd", "kitti_3classes", "scannet_large", "nuscenes")


def _strip(cfg):
    cfg = dict(cfg)
    cfg.pop("type")
    return rs.to_attr(cfg)


def gen_dense_stack(nd):
    """SECOND3D (models/backbones/second_3d.py:52-76,89-114) + SECOND3DFPN (models/necks/second3d_fpn.py:48-104,112-143) built from
    the shipped configs' own pts_backbone / pts_neck dicts, training-mode BatchNorm, name-seeded weights, a small lattice:
    the three backbone outputs, the neck output, the gradient of a seeded linear functional of the neck output w.r.t. the input,
    and the state-dict names / shapes.  Configs whose two dicts equal SUN RGB-D's are recorded as aliases, not stored again."""
    out, seen = {}, {}
    for name in DENSE_CFGS:
        m = rs.model_cfg(name)
        sig = repr((m["pts_backbone"], m["pts_neck"]))
        if sig in seen:
            out[f"{name}.same_as"] = np.array(seen[sig])
            continue
        seen[sig] = name
        bb = nd.backbone.SECOND3D(**_strip(m["pts_backbone"]))
        nk = nd.neck.SECOND3DFPN(**_strip(m["pts_neck"]))
        sd_b = seeded_state_dict(bb.state_dict().items(), SEED)
        sd_n = seeded_state_dict(nk.state_dict().items(), SEED)
        bb.load_state_dict(sd_b)
        nk.load_state_dict(sd_n)
        bb.train()
        nk.train()
        cin = m["pts_backbone"]["in_channels"][0]
        x = seeded_input(f"dense_in.{name}", (2, cin, 2, 8, 8), SEED, -0.5, 1.0).clamp_min(0).requires_grad_(True)
        feats = bb(x)
        y = nk(feats)
        cot = seeded_input(f"dense_cot.{name}", tuple(y.shape), SEED, -1.0, 1.0)
        (y * cot).sum().backward()
        out[f"{name}.x"] = x.detach().numpy()
        for i, f in enumerate(feats):
            out[f"{name}.backbone{i}"] = f.detach().numpy()
        out[f"{name}.neck"] = y.detach().numpy()
        out[f"{name}.cot"] = cot.numpy()
        out[f"{name}.dx"] = x.grad.numpy()
        # parameter gradients, kept small: every BatchNorm scale / shift gradient of both modules (each depends on the whole
        # backward chain behind it) and slices of two convolution weight gradients (first conv of branch 0; the stride-4 transposed conv)
        out[f"{name}.wgrad_first_8"] = bb.blocks[0][0].weight.grad[:8].numpy().copy()
        out[f"{name}.wgrad_deconv2_4"] = nk.deblocks[2][0].weight.grad[:, :4].numpy().copy()
        bn_keys = [k for k, p_ in list(bb.named_parameters()) + list(nk.named_parameters()) if p_.dim() == 1]
        out[f"{name}.bn_grad_keys"] = np.array(["pts_backbone." + k for k, _ in bb.named_parameters() if _.dim() == 1]
                                               + ["pts_neck." + k for k, _ in nk.named_parameters() if _.dim() == 1])
        out[f"{name}.bn_grads"] = np.concatenate([p_.grad.numpy().reshape(-1) for p_ in list(bb.parameters()) + list(nk.parameters())
                                                  if p_.dim() == 1])
        assert len(bn_keys) == len(out[f"{name}.bn_grad_keys"])
        out[f"{name}.backbone_keys"] = np.array(list(sd_b.keys()))
        out[f"{name}.backbone_shapes"] = np.array([repr(tuple(v.shape)) for v in sd_b.values()])
        out[f"{name}.neck_keys"] = np.array(list(sd_n.keys()))
        out[f"{name}.neck_shapes"] = np.array([repr(tuple(v.shape)) for v in sd_n.values()])
        print("dense_stack", name, tuple(y.shape), float(y.detach().abs().mean()))
    np.savez_compressed(os.path.join(OUT, "dense_stack.npz"), seed=SEED, **out)


def _wiring_scene(name, cin, shape, B, n):
    """A seeded sparse input on a small grid: clustered active sites (so that strided levels merge neighbours) + features."""
    rng = np.random.default_rng([SEED, sum(map(ord, name))])
    cs = []
    for b in range(B):
        ctr = rng.integers(0, [shape[0], shape[1], shape[2]], (6, 3))
        pts = ctr[rng.integers(0, 6, n)] + rng.integers(-3, 4, (n, 3))
        pts = pts[np.all((pts >= 0) & (pts < np.array(shape)), axis=1)]
        pts = np.unique(pts, axis=0)
        pts = pts[rng.permutation(pts.shape[0])]                        # voxel order is arbitrary upstream
        cs.append(np.concatenate([np.full((pts.shape[0], 1), b), pts], 1))
    coors = np.concatenate(cs).astype(np.int32)
    feats = rng.uniform(-1, 1, (coors.shape[0], cin)).astype(np.float32)
    return feats, coors


def gen_encoder_wiring(nd):
    """SparseEncoderHD.__init__ / make_encoder_layers / forward (models/pts_encoder/sparse_encoder_hd.py:36-138, 140-214) built from the
    shipped configs' own pts_middle_encoder dicts (only `sparse_shape` is replaced by a small grid) over refshim's stand-in sparse
    layers (= conv3d on the densified tensor, read back on the active set): the sequence of sparse-conv calls (kind, channels,
    kernel, stride, padding, indice_key, active rows in / out), the dense output, and the state-dict names / shapes."""
    out = {}
    shape, B = (24, 32, 32), 2
    for name in DENSE_CFGS:
        m = rs.model_cfg(name)
        ec = dict(m["pts_middle_encoder"])
        ec["sparse_shape"] = list(shape)
        enc = nd.encoder.SparseEncoderHD(**_strip(ec))
        sd = seeded_state_dict(enc.state_dict().items(), SEED)
        enc.load_state_dict(sd)
        enc.train()
        feats, coors = _wiring_scene(name, ec["in_channels"], shape, B, 220)
        del rs.CONV_TRACE[:]
        with torch.no_grad():
            y = enc(torch.from_numpy(feats), torch.from_numpy(coors), B)
        tr = list(rs.CONV_TRACE)
        out[f"{name}.feats"], out[f"{name}.coors"] = feats, coors
        out[f"{name}.dense"] = y.numpy()
        out[f"{name}.trace_kind"] = np.array([t["kind"] for t in tr])
        out[f"{name}.trace_key"] = np.array([t["indice_key"] for t in tr])
        out[f"{name}.trace_num"] = np.array([[t["cin"], t["cout"], *t["kernel"], *t["stride"], *t["padding"], t["n_in"], t["n_out"],
                                              *t["shape_in"], *t["shape_out"]] for t in tr], np.int64)
        out[f"{name}.keys"] = np.array(list(sd.keys()))
        out[f"{name}.shapes"] = np.array([repr(tuple(v.shape)) for v in sd.values()])
        print("encoder_wiring", name, len(tr), "sparse convs,", tuple(y.shape), "active rows", [t["n_out"] for t in tr][::5])
    np.savez_compressed(os.path.join(OUT, "encoder_wiring.npz"), seed=SEED, sparse_shape=np.array(shape), **out)


def gen_detector_glue(nd):
    """models/detectors/uni3detr.py:18-46 `shift_scale_points`, as the detector calls it (:181,:187: dst_range None, src_range =
    per-scene min / max) plus the general form (explicit dst_range; the 4-D branch :31-33)."""
    rng = np.random.default_rng(SEED + 11)
    f = nd.detector.shift_scale_points
    x3 = torch.from_numpy(rng.uniform(-4, 7, (3, 50, 3)).astype(np.float32))
    src3 = [x3.min(dim=1)[0], x3.max(dim=1)[0]]
    y_unit = f(x3, src_range=src3)
    dst = [torch.from_numpy(rng.uniform(-1, 0, (3, 3)).astype(np.float32)), torch.from_numpy(rng.uniform(1, 2, (3, 3)).astype(np.float32))]
    y_dst = f(x3, src_range=src3, dst_range=dst)
    ints = torch.from_numpy(rng.integers(0, 320, (2, 300, 3)).astype(np.float32))      # float-cast voxel coordinates (:183-187)
    srci = [ints.min(dim=1)[0], ints.max(dim=1)[0]]
    y_int = f(ints, src_range=srci)
    x4 = torch.from_numpy(rng.uniform(-2, 2, (2, 4, 9, 3)).astype(np.float32))
    src4 = [x4.reshape(2, -1, 3).min(dim=1)[0], x4.reshape(2, -1, 3).max(dim=1)[0]]
    y4 = f(x4, src_range=src4)
    np.savez_compressed(os.path.join(OUT, "detector_glue.npz"), x3=x3.numpy(), y_unit=y_unit.numpy(), dst_lo=dst[0].numpy(),
                        dst_hi=dst[1].numpy(), y_dst=y_dst.numpy(), ints=ints.numpy(), y_int=y_int.numpy(), x4=x4.numpy(), y4=y4.numpy())
    print("detector_glue ok", tuple(y4.shape))


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
    nd = rs.load_dense_path()
    gen_dense_stack(nd)
    gen_encoder_wiring(nd)
    gen_detector_glue(nd)


if __name__ == "__main__":
    main()
