"""ORACLE (test infrastructure, never imported by the product path; also the `cpu_baseline` "port" leg of bench.py).

Functional pure-torch-CPU restatement of the whole Uni3DETR training hot path:
voxelize -> VFE -> SparseEncoderHD -> SECOND3D -> SECOND3DFPN -> 2x FPS -> Uni3DETRHead/Transformer ->
HungarianAssigner3D -> losses.  Parameters come in as a flat state_dict with the reference's key names
(SURVEY.md Appendix C).  Every function cites the reference file:line it follows.

Pinning: everything the reference itself implements is checked against golden vectors generated from the reference's own
files (oracle/make_golden.py -> tests/golden/*.npz; tests/test_oracle_cpu.py): decoder / head / matcher / losses / coder, and
(round 6) SECOND3D + SECOND3DFPN (dense_stack.npz), SparseEncoderHD's layer wiring (encoder_wiring.npz) and
shift_scale_points (detector_glue.npz).  The voxelize / sparse-conv arithmetic / BN1d / FPS / IoU pieces restate un-vendored
upstream ops (mmcv, mmdet3d, spconv): PARITY UNPINNED there (property tests only).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import boxes as ob
from . import geometry as og


# ==================================================================================================
# config (the numbers of projects/configs/uni3detr/uni3detr_sunrgbd.py:9-140, restated)
# ==================================================================================================
def sunrgbd_cfg():
    return dict(
        voxel_size=(0.02, 0.02, 0.02), pc_range=(-3.2, -0.2, -2.0, 3.2, 6.2, 0.56), sparse_shape=(128, 320, 320),
        max_points=5, max_voxels=(16000, 40000), num_features=4,
        enc_in=4, enc_base=16, enc_out=256,
        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
        encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (0, 1, 1)), (0, 0)), encoder_strides=(2, 2, 2, 1),
        bb_in=(256, 256, 256), bb_out=(128, 256, 512), bb_layers=(5, 5, 5), bb_strides=(1, 2, 4), bb_kernel=(1, 3, 3),
        fpn_in=(128, 256, 512), fpn_out=(256, 256, 256), fpn_strides=(1, 2, 4), fpn_extra=3,
        bn_eps=1e-3, num_query=300, num_classes=10, code_size=8, embed=256, heads=8, ffn=512, dec_layers=3,
        cls_w=1.5, bbox_w=0.25, iou_w=1.2, cost_cls=2.0, cost_reg=0.25, cost_iou=1.2, alpha=0.25, gamma=2.0,
        code_weights=(1.0,) * 8, fps_packed_quirk=True)


def _derive(**kw):
    c = sunrgbd_cfg()
    c.update(kw)
    return c


def kitti_cfg():
    """projects/configs/uni3detr/uni3detr_kitti_3classes.py:10-11,28-41,64-77: outdoor range, 0.05/0.05/0.1 voxels, 9 decoder layers."""
    return _derive(voxel_size=(0.05, 0.05, 0.1), pc_range=(0, -40, -3, 70.4, 40, 1), sparse_shape=(41, 1600, 1408), num_classes=3,
                   dec_layers=9)


def scannet_large_cfg():
    """uni3detr_scannet_large.py:9-12,28-42,64-70: dynamic voxelization, base_channels 32, 512-channel encoder output, 18 classes."""
    return _derive(pc_range=(-6.4, -6.4, -0.1, 6.4, 6.4, 2.46), sparse_shape=(128, 640, 640), dynamic=True, enc_base=32, enc_out=512,
                   encoder_channels=((32, 32, 64), (64, 64, 128), (128, 128, 256), (256, 256)), bb_in=(512, 512, 512), num_classes=18)


def nuscenes_cfg():
    """uni3detr_nuscenes.py:13-14,31-46,69: 5-feature points, 10 points / voxel, 900 queries, code size 10 (the head's default)."""
    return _derive(voxel_size=(0.075, 0.075, 0.2), pc_range=(-54, -54, -5.0, 54, 54, 3.0), sparse_shape=(41, 1440, 1440), max_points=10,
                   max_voxels=(90000, 120000), num_features=5, enc_in=5, num_query=900, code_size=10, code_weights=(1.0,) * 10)


# ==================================================================================================
# a-4  SparseEncoderHD  (models/pts_encoder/sparse_encoder_hd.py:106-138, layer list :140-214)
# ==================================================================================================
def _bn_rows(sd, key, x, eps, residual=None, relu=True):
    return og.bn_train(x, sd[key + ".weight"], sd[key + ".bias"], eps, residual, relu)


def _w27(w):
    return w.reshape(-1, w.shape[-2], w.shape[-1])


def sparse_encoder(sd, pre, feats, coors, batch, cfg):
    """feats [N,Cin] f32, coors [N,4] (b,z,y,x) numpy -> dense [B,Cout,D,H,W] (keep_depth=True)."""
    k3, s1, p1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    dims = tuple(cfg["sparse_shape"])
    coors = np.asarray(coors)
    eps = cfg["bn_eps"]

    def subm(x, wkey, c, d):
        return og.sparse_conv(x, _w27(sd[wkey]), og.nbr_table(c, c, d, k3, s1, p1, 0))

    # conv_input: SubMConv3d(k3,p1) + BN1d + ReLU (:80-88)
    x = _bn_rows(sd, pre + "conv_input.1", subm(feats, pre + "conv_input.0.weight", coors, dims), eps)
    for i, blocks in enumerate(cfg["encoder_channels"]):
        st = pre + f"encoder_layers.encoder_layer{i + 1}."
        last_stage = i == len(cfg["encoder_channels"]) - 1
        for j in range(len(blocks)):
            if j == len(blocks) - 1 and not last_stage:
                # SparseConv3d k3 stride s pad p + BN + ReLU (:183-192)
                pad = cfg["encoder_paddings"][i][j]
                pad = tuple(pad) if isinstance(pad, (tuple, list)) else (pad,) * 3
                s = (cfg["encoder_strides"][i],) * 3
                oc, odims = og.strided_out_coords(coors, dims, k3, s, pad)
                nbr = og.nbr_table(oc, coors, dims, k3, s, pad, 0)
                x = _bn_rows(sd, st + f"{j}.1", og.sparse_conv(x, _w27(sd[st + f"{j}.0.weight"]), nbr), eps)
                coors, dims = oc, odims
            else:
                # SparseBasicBlock (:195-199; upstream mmdet3d)
                idn = x
                o = _bn_rows(sd, st + f"{j}.bn1", subm(x, st + f"{j}.conv1.weight", coors, dims), eps)
                x = _bn_rows(sd, st + f"{j}.bn2", subm(o, st + f"{j}.conv2.weight", coors, dims), eps, residual=idn)
    # conv_out 1x1x1 + BN + ReLU (:96-104), dense() (:133)
    x = _bn_rows(sd, pre + "conv_out.1", x @ _w27(sd[pre + "conv_out.0.weight"])[0], eps)
    return og.to_dense(x, coors, batch, dims)


# ==================================================================================================
# a-10  SECOND3D / SECOND3DFPN  (models/backbones/second_3d.py:52-76,89-114; models/necks/second3d_fpn.py:48-143)
# ==================================================================================================
def _bn3(sd, key, x, eps):
    return F.relu(F.batch_norm(x, None, None, sd[key + ".weight"], sd[key + ".bias"], True, 0.0, eps))


def second3d(sd, pre, x, cfg):
    outs = []
    k = cfg["bb_kernel"]
    pad = tuple((kk - 1) // 2 for kk in k)
    for i, (n, s) in enumerate(zip(cfg["bb_layers"], cfg["bb_strides"])):
        y = x                                                       # is_cascade=False: every block sees the input
        for j in range(n + 1):
            stride = (1, s, s) if j == 0 else (1, 1, 1)
            y = F.conv3d(y, sd[pre + f"blocks.{i}.{3 * j}.weight"], None, stride, pad)
            y = _bn3(sd, pre + f"blocks.{i}.{3 * j + 1}", y, cfg["bn_eps"])
        outs.append(y)
    return outs


def second3dfpn(sd, pre, xs, cfg):
    ups = []
    for i, s in enumerate(cfg["fpn_strides"]):
        w = sd[pre + f"deblocks.{i}.0.weight"]
        if s > 1:
            y = F.conv_transpose3d(xs[i], w, None, (1, s, s))
        else:
            y = F.conv3d(xs[i], w, None, (1, 1, 1))
        ups.append(_bn3(sd, pre + f"deblocks.{i}.1", y, cfg["bn_eps"]))
    out = ups[0]
    for u in ups[1:]:
        out = out + u
    for j in range(cfg["fpn_extra"]):
        out = F.conv3d(out, sd[pre + f"extra_blocks.{3 * j}.weight"], None, 1, 1)
        out = _bn3(sd, pre + f"extra_blocks.{3 * j + 1}", out, cfg["bn_eps"])
    return out


# ==================================================================================================
# a-3  FPS (models/detectors/uni3detr.py:138,178-189; upstream mmcv furthest_point_sample)
# ==================================================================================================
def fps_packed(flat, n, m):
    """D-FPS exactly as the upstream kernel reads memory: point k = flat[3k:3k+3]; start index 0; running min of
    squared L2 (float32, left-to-right sum); argmax tie rule of the upstream block reduction: among equal maxima
    the winner has the smallest (k mod T, k), T = min(1024, 2^floor(log2 n))."""
    p = np.asarray(flat, np.float32).reshape(-1)[: 3 * n].reshape(n, 3)
    T = max(min(1 << int(math.floor(math.log2(n))), 1024), 1)
    tie = (np.arange(n) % T).astype(np.int64) * n + np.arange(n)
    mind = np.full(n, 1e10, np.float32)
    idx = np.zeros(m, np.int64)
    old = 0
    for j in range(1, m):
        d = p - p[old]
        dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        mind = np.minimum(mind, dist.astype(np.float32))
        best = mind.max()
        cand = np.nonzero(mind == best)[0]
        old = int(cand[np.argmin(tie[cand])])
        idx[j] = old
    return idx


def shift_scale_points(x, src_lo, src_hi, dst_lo=None, dst_hi=None):
    """models/detectors/uni3detr.py:18-46: affine map of x [B,N,3] (or [B,M,N,3]: ranges gain an axis, :31-33) from the per-scene box
    [src_lo, src_hi] ([B,3]) onto [dst_lo, dst_hi] (default the unit cube, :24-28): ((x - src_lo) * dst_diff) / src_diff + dst_lo, in
    that operation order (:42-45).  Pinned by tests/golden/detector_glue.npz."""
    if dst_lo is None:
        dst_lo, dst_hi = torch.zeros_like(src_lo), torch.ones_like(src_lo)
    if x.dim() == 4:
        src_lo, src_hi, dst_lo, dst_hi = (t[:, None] for t in (src_lo, src_hi, dst_lo, dst_hi))
    src_diff = src_hi[:, None, :] - src_lo[:, None, :]
    dst_diff = dst_hi[:, None, :] - dst_lo[:, None, :]
    return ((x - src_lo[:, None, :]) * dst_diff) / src_diff + dst_lo[:, None, :]


def shift_scale_unit(x):
    """shift_scale_points(x, src=[min,max]) with dst=[0,1] (uni3detr.py:18-46,:181)."""
    return shift_scale_points(x, x.min(dim=1)[0], x.max(dim=1)[0])


def fps_queries(points_list, coors, cfg):
    """-> fpsbpts [B,600,3] in [0,1] (uni3detr.py:178-189)."""
    m = cfg["num_query"]
    a, b = [], []
    for i, pts in enumerate(points_list):
        pts = np.asarray(pts, np.float32)
        flat = pts.reshape(-1) if cfg.get("fps_packed_quirk", True) else np.ascontiguousarray(pts[:, :3]).reshape(-1)
        idx = fps_packed(flat, pts.shape[0], m)
        a.append(torch.from_numpy(pts[idx, :3]))
        vc = coors[coors[:, 0] == i][:, 1:].astype(np.float32)            # (z,y,x) voxel coords of scene i
        idx2 = fps_packed(vc.reshape(-1), vc.shape[0], m)
        b.append(torch.from_numpy(vc[idx2][:, [2, 1, 0]]))
    fa = shift_scale_unit(torch.stack(a))
    fb = shift_scale_unit(torch.stack(b))
    return torch.cat([fa, fb], 1)


# ==================================================================================================
# a-5/a-6/a-7  head + transformer  (dense_heads/uni3detr_head.py:422-508; utils/uni3detr_transformer.py:33-360)
# ==================================================================================================
def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def sine_embed(pos, num_feats=128, temperature=10000):
    """get_sine_pos_embed (uni3detr_transformer.py:33-65): pos [B,N,3] -> [B,N,384]."""
    dim_t = torch.arange(num_feats, dtype=pos.dtype)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_feats)
    res = []
    for i in range(pos.shape[-1]):
        s = pos[..., i:i + 1] * (2 * math.pi) / dim_t
        res.append(torch.stack((s[..., 0::2].sin(), s[..., 1::2].cos()), dim=3).flatten(2))
    return torch.cat(res, dim=2)


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def _mlp3(sd, key, x):
    x = F.relu(_lin(sd, key + ".layers.0", x))
    x = F.relu(_lin(sd, key + ".layers.1", x))
    return _lin(sd, key + ".layers.2", x)


def _ln(sd, key, x):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"])


def _self_attn(sd, key, q, pos, heads):
    """mmcv MultiheadAttention wrapper, batch-first [B,N,C] here; dropout off."""
    B, N, C = q.shape
    w, b = sd[key + ".attn.in_proj_weight"], sd[key + ".attn.in_proj_bias"]
    qk = q + pos
    Q = F.linear(qk, w[:C], b[:C]).view(B, N, heads, C // heads).transpose(1, 2)
    K = F.linear(qk, w[C:2 * C], b[C:2 * C]).view(B, N, heads, C // heads).transpose(1, 2)
    V = F.linear(q, w[2 * C:], b[2 * C:]).view(B, N, heads, C // heads).transpose(1, 2)
    att = torch.softmax(Q @ K.transpose(-1, -2) / math.sqrt(C // heads), dim=-1)
    o = (att @ V).transpose(1, 2).reshape(B, N, C)
    return q + F.linear(o, sd[key + ".attn.out_proj.weight"], sd[key + ".attn.out_proj.bias"])


def _cross_attn(sd, key, q, pos, ref_logits, value):
    """UniCrossAtten.forward (uni3detr_transformer.py:271-360), value [B,C,D,H,W]."""
    w = torch.sigmoid(_lin(sd, key + ".attention_weights", q + pos))                 # [B,N,1]
    grid = ((torch.sigmoid(ref_logits) - 0.5) * 2).view(q.shape[0], 1, 1, -1, 3)
    samp = F.grid_sample(value, grid, mode="bilinear", padding_mode="zeros", align_corners=False)  # [B,C,1,1,N]
    samp = samp.view(q.shape[0], value.shape[1], -1).transpose(1, 2)                  # [B,N,C]
    out = _lin(sd, key + ".output_proj", samp * w)
    pe = F.relu(_ln(sd, key + ".position_encoder.1", _lin(sd, key + ".position_encoder.0", ref_logits)))
    pe = F.relu(_ln(sd, key + ".position_encoder.4", _lin(sd, key + ".position_encoder.3", pe)))
    return out + q + pe


def _ffn(sd, key, x):
    return x + _lin(sd, key + ".layers.1", F.relu(_lin(sd, key + ".layers.0.0", x)))


def decoder_group(sd, pre, query, ref_logits, value, cfg):
    """Uni3DETRTransformerDecoder.forward for one query group (uni3detr_transformer.py:145-212), batch-first."""
    dec = pre + "transformer.decoder."
    states, refs = [], []
    out = query
    for lid in range(cfg["dec_layers"]):
        raw = _mlp3(sd, dec + "ref_point_head", sine_embed(torch.sigmoid(ref_logits)))
        pos = raw if lid == 0 else _mlp3(sd, dec + "query_scale", out) * raw
        L = dec + f"layers.{lid}."
        out = _ln(sd, L + "norms.0", _self_attn(sd, L + "attentions.0", out, pos, cfg["heads"]))
        out = _ln(sd, L + "norms.1", _cross_attn(sd, L + "attentions.1", out, pos, ref_logits, value))
        out = _ln(sd, L + "norms.2", _ffn(sd, L + "ffns.0", out))
        tmp = _branch(sd, pre + f"reg_branches.{lid}", out, ln=False)
        new_ref = torch.cat([tmp[..., :2] + ref_logits[..., :2], tmp[..., 4:5] + ref_logits[..., 2:3]], -1)
        ref_logits = new_ref.detach()
        states.append(out)
        refs.append(ref_logits)
    return states, refs


def _branch(sd, key, x, ln):
    if ln:   # cls branch: Linear-LN-ReLU x2 + Linear (uni3detr_head.py:367-373)
        x = F.relu(_ln(sd, key + ".1", _lin(sd, key + ".0", x)))
        x = F.relu(_ln(sd, key + ".4", _lin(sd, key + ".3", x)))
        return _lin(sd, key + ".6", x)
    x = F.relu(_lin(sd, key + ".0", x))       # reg / iou branch (:375-387)
    x = F.relu(_lin(sd, key + ".2", x))
    return _lin(sd, key + ".4", x)


def head_forward(sd, pre, pts_feats, fpsbpts, cfg, rand_points=None):
    """Uni3DETRHead.forward (uni3detr_head.py:422-508).  Train layout (3 groups) unless rand_points is given (eval, 4 groups)."""
    nq = cfg["num_query"]
    B = fpsbpts.shape[0]
    tgt = sd[pre + "tgt_embed.weight"]
    anchor = sd[pre + "refpoint_embed.weight"]
    groups_t = [tgt[:nq], tgt[nq:], tgt[nq:]]
    groups_r = [anchor[None].expand(B, -1, -1), inverse_sigmoid(fpsbpts[:, :nq]), inverse_sigmoid(fpsbpts[:, nq:])]
    if rand_points is not None:
        groups_t.append(tgt[nq:])
        groups_r.append(inverse_sigmoid(rand_points))
    pr = cfg["pc_range"]
    cls_l = [[] for _ in range(cfg["dec_layers"])]
    box_l = [[] for _ in range(cfg["dec_layers"])]
    iou_l = [[] for _ in range(cfg["dec_layers"])]
    for t, r in zip(groups_t, groups_r):
        states, refs = decoder_group(sd, pre, t[None].expand(B, -1, -1), r, pts_feats, cfg)
        for l in range(cfg["dec_layers"]):
            # reference = inverse_sigmoid(sigmoid(ref)) of the *input* ref of layer l (:463-468)
            ref_in = r if l == 0 else refs[l - 1]
            ref = inverse_sigmoid(torch.sigmoid(ref_in))
            hs = states[l]
            cls_l[l].append(_branch(sd, pre + f"cls_branches.{l}", hs, ln=True))
            tmp = _branch(sd, pre + f"reg_branches.{l}", hs, ln=False)
            xy = torch.sigmoid(tmp[..., 0:2] + ref[..., 0:2])
            z = torch.sigmoid(tmp[..., 4:5] + ref[..., 2:3])
            x_ = xy[..., 0:1] * (pr[3] - pr[0]) + pr[0]
            y_ = xy[..., 1:2] * (pr[4] - pr[1]) + pr[1]
            z_ = z * (pr[5] - pr[2]) + pr[2]
            box_l[l].append(torch.cat([x_, y_, tmp[..., 2:4], z_, tmp[..., 5:]], -1))
            iou_l[l].append(_branch(sd, pre + f"iou_branches.{l}", hs, ln=False))
    cat = lambda L: torch.stack([torch.cat(x, 1) for x in L])
    return cat(cls_l), cat(box_l), cat(iou_l)


# ==================================================================================================
# a-8/a-9  assigner + losses  (core/bbox/assigners/hungarian_assigner_3d.py:53-151; uni3detr_head.py:510-793;
#          core/bbox/util.py:8-80; models/losses/rdiouloss.py:93-102,162-184)
# ==================================================================================================
def normalize_bbox(b):
    rot = -b[..., 6:7] - math.pi / 2
    cols = [b[..., 0:1], b[..., 1:2], (b[..., 4:5] + 1e-5).log(), (b[..., 3:4] + 1e-5).log(), b[..., 2:3],
            (b[..., 5:6] + 1e-5).log(), rot.sin(), rot.cos()]
    if b.shape[-1] > 7:                      # velocity pair appended as is (core/bbox/util.py:32-37)
        cols += [b[..., 7:8], b[..., 8:9]]
    return torch.cat(cols, -1)


def denormalize_bbox(n):
    rot = -torch.atan2(n[..., 6:7], n[..., 7:8]) - math.pi / 2
    return torch.cat([n[..., 0:1], n[..., 1:2], n[..., 4:5], n[..., 3:4].exp(), n[..., 2:3].exp(), n[..., 5:6].exp(), rot], -1)


def match_cost(cls_pred, bbox_pred, gt, labels, cfg):
    """cost [Q,G] = focal + L1(cdist p=1 on 8 code dims) + (1 - nearestBEV IoU) with the config weights."""
    p = cls_pred.sigmoid()
    a, g = cfg["alpha"], cfg["gamma"]
    neg = -(1 - p + 1e-12).log() * (1 - a) * p.pow(g)
    pos = -(p + 1e-12).log() * a * (1 - p).pow(g)
    c_cls = (pos[:, labels] - neg[:, labels]) * cfg["cost_cls"]
    c_reg = torch.cdist(bbox_pred[:, :8], normalize_bbox(gt[:, :7])[:, :8], p=1) * cfg["cost_reg"]
    c_iou = (1 - ob.bbox_overlaps_nearest_3d(denormalize_bbox(bbox_pred), gt[:, :7])) * cfg["cost_iou"]
    return c_cls + c_reg + c_iou


def assign(cls_pred, bbox_pred, gt, labels, cfg):
    """-> assigned gt index per query (0 = background, 1-based), per 300-query group LSA (gt_repeattimes is always 1:
    SURVEY.md App. D-1)."""
    from scipy.optimize import linear_sum_assignment
    Q = bbox_pred.shape[0]
    out = torch.zeros(Q, dtype=torch.long)
    if gt.shape[0] == 0:
        return out
    cost = match_cost(cls_pred, bbox_pred, gt, labels, cfg).detach().cpu()
    nq = cfg["num_query"]
    for g in range(Q // nq):
        r, c = linear_sum_assignment(cost[g * nq:(g + 1) * nq])
        out[torch.from_numpy(g * nq + r)] = torch.from_numpy(c) + 1
    return out


def loss_single(cls, box, iou_pred, gts, labels, cfg, assigned=None):
    """loss_single (uni3detr_head.py:617-698) for one decoder layer; world size 1.
    assigned: optional [B,Q] matching to use INSTEAD of running the assigner (checker-only knob: gradient parity of a reduced-precision
    product run is measured under the product's own discrete matching, so that what is compared is backward arithmetic).
    Target width: 7 columns, or 9 when the box code has 10 entries (nuScenes) - the reference's `[..., :7]` slice (:557) cannot
    feed its own 10-column L1 (:684-687), so uni3detr_nuscenes.py does not train as shipped; this follows the commented-out upstream
    `[..., :9]` line above it, with zero velocities for 7-column GT (same choice as the product: plugin/head.py gt_dim)."""
    B, Q, C = cls.shape
    lab_t = torch.full((B * Q,), C, dtype=torch.long)
    gd = 9 if box.shape[-1] >= 10 else 7
    gts = [F.pad(g[:, :gd], (0, gd - min(gd, g.shape[1]))).to(box.dtype) for g in gts]
    tgt = torch.zeros(B * Q, gd, dtype=box.dtype)
    wgt = torch.zeros(B * Q, box.shape[-1], dtype=box.dtype)
    npos = 0
    assigned_in, assigned = assigned, []
    for b in range(B):
        a = assign(cls[b], box[b], gts[b], labels[b], cfg) if assigned_in is None else assigned_in[b].long()
        assigned.append(a)
        posm = a > 0
        idx = torch.nonzero(posm).squeeze(-1)
        lab_t[b * Q + idx] = labels[b][a[idx] - 1]
        tgt[b * Q + idx] = gts[b][a[idx] - 1]
        wgt[b * Q + idx] = 1.0
        npos += int(posm.sum())
    cls_avg = max(float(npos), 1.0)
    cls = cls.reshape(-1, C)
    box = box.reshape(-1, box.shape[-1])
    ntgt = normalize_bbox(tgt)
    b3d = denormalize_bbox(box)
    iou_bev = ob.bbox_overlaps_nearest_3d(b3d, tgt[:, :7], is_aligned=True)
    z1, z2 = b3d[:, 2] - b3d[:, 5] / 2, b3d[:, 2] + b3d[:, 5] / 2
    z3, z4 = tgt[:, 2] - tgt[:, 5] / 2, tgt[:, 2] + tgt[:, 5] / 2
    iou_z = torch.max(torch.min(z2, z4) - torch.max(z1, z3), torch.zeros_like(z1)) / (torch.max(z2, z4) - torch.min(z1, z3))
    q = (iou_bev + iou_z) / 2                                                # NOT detached (App. D-6)
    # soft focal loss (rdiouloss.py:162-184), label weights = 1
    ps = cls.sigmoid()
    oh = F.one_hot(lab_t, C + 1)[:, :C].to(cls.dtype)
    ts = oh * q[:, None]
    pt = ts - ps
    fw = ((1 - cfg["alpha"]) + (2 * cfg["alpha"] - 1) * ts) * pt.pow(cfg["gamma"])
    l_cls = (F.binary_cross_entropy_with_logits(cls, ts, reduction="none") * fw).sum() / (cls_avg + torch.finfo(torch.float32).eps)
    l_cls = cfg["cls_w"] * l_cls
    npos_c = max(float(npos), 1.0)
    cw = torch.tensor(cfg["code_weights"], dtype=box.dtype)
    wgt = wgt * cw
    eps32 = torch.finfo(torch.float32).eps
    l_box = cfg["bbox_w"] * ((box - ntgt).abs() * wgt).sum() / (npos_c + eps32)
    if bool((wgt > 0).any()):
        l_iou = cfg["iou_w"] * ((1 - iou_bev) * wgt.mean(-1)).sum() / (npos_c + eps32)
    else:
        l_iou = b3d.sum() * wgt.sum()
    l_iou = l_iou + ((1 - iou_z) * wgt[:, 0]).sum() / npos_c
    iou_true = ob.bbox_overlaps_3d_aligned(b3d.detach(), tgt[:, :7]).to(iou_pred.dtype)
    l_ioup = (F.binary_cross_entropy_with_logits(iou_pred.reshape(-1), iou_true, reduction="none") * wgt[:, 0]).sum() / npos_c * 1.2
    return (l_cls, l_box, l_iou, l_ioup), assigned


def head_loss(cls_all, box_all, iou_all, gts_bottom, labels, cfg, assigned=None):
    """Uni3DETRHead.loss (uni3detr_head.py:716-793).  gts_bottom: list of [G,7] with bottom-centre z (box `.tensor`);
    converted to gravity centre as :759-761."""
    gts = [torch.cat([g[:, :2], g[:, 2:3] + g[:, 5:6] * 0.5, g[:, 3:]], 1) for g in gts_bottom]
    L = cls_all.shape[0]
    assigned_in, out, assigned = assigned, {}, []
    for l in range(L):
        (lc, lb, li, lp), a = loss_single(cls_all[l], box_all[l], iou_all[l], gts, labels, cfg, None if assigned_in is None else assigned_in[l])
        pfx = "" if l == L - 1 else f"d{l}."
        out[pfx + "loss_cls"], out[pfx + "loss_bbox"], out[pfx + "loss_iou"], out[pfx + "loss_iou_pred"] = lc, lb, li, lp
        assigned.append(torch.stack(a))
    return out, torch.stack(assigned)


# ==================================================================================================
# a-1  whole training forward (models/detectors/uni3detr.py:143-266)
# ==================================================================================================
def forward_features(sd, points_list, cfg):
    """extract_pts_feat (uni3detr.py:143-190) for either voxelization mode -> (features [B,C,D,H,W], fpsbpts [B,2*nq,3])."""
    B = len(points_list)
    if cfg.get("dynamic"):
        # :155-171: per-point coors (-1 rows kept), DynamicSimpleVFE mean; the voxel-coordinate FPS then runs over the PER-POINT coors
        pcoors, feats, coors = og.voxelize_dynamic(points_list, cfg["voxel_size"], cfg["pc_range"])
        feats = torch.from_numpy(feats[:, :cfg["num_features"]])
        fps_coors = pcoors
    else:
        vox, coors, num = og.voxelize_batch(points_list, cfg["voxel_size"], cfg["pc_range"], cfg["max_points"], cfg["max_voxels"][0])
        feats = torch.from_numpy(og.vfe_mean(vox, num, cfg["num_features"]))
        fps_coors = coors
    x = sparse_encoder(sd, "pts_middle_encoder.", feats, coors, B, cfg)
    x = second3dfpn(sd, "pts_neck.", second3d(sd, "pts_backbone.", x, cfg), cfg)
    return x, fps_queries(points_list, fps_coors, cfg)


def forward_logits(sd, points_list, cfg):
    """features + head outputs, no loss (nuScenes: the reference's loss path is inconsistent with its 9-dim GT)."""
    x, fpsbpts = forward_features(sd, points_list, cfg)
    cls, box, iou = head_forward(sd, "pts_bbox_head.", x, fpsbpts, cfg)
    return dict(cls=cls, box=box, iou=iou, fpsbpts=fpsbpts, feats=x)


def forward_train(sd, points_list, gts_bottom, labels, cfg, assigned=None):
    """assigned: optional [L,B,Q] matching override (see loss_single).  The arithmetic runs in the dtype of `sd` (float32 as the
    reference; float64 = the checker's conditioning yardstick, tests/test_grad_parity_gpu.py): voxelization, VFE means and FPS are
    float32 either way (they define the integer results) and are cast at the boundary."""
    B = len(points_list)
    dt = sd["pts_bbox_head.tgt_embed.weight"].dtype
    vox, coors, num = og.voxelize_batch(points_list, cfg["voxel_size"], cfg["pc_range"], cfg["max_points"], cfg["max_voxels"][0])
    feats = torch.from_numpy(og.vfe_mean(vox, num, cfg["num_features"])).to(dt)
    enc = sparse_encoder(sd, "pts_middle_encoder.", feats, coors, B, cfg)
    bb = second3d(sd, "pts_backbone.", enc, cfg)
    x = second3dfpn(sd, "pts_neck.", bb, cfg)
    fpsbpts = fps_queries(points_list, coors, cfg).to(dt)
    gts_bottom = [g.to(dt) for g in gts_bottom]
    cls, box, iou = head_forward(sd, "pts_bbox_head.", x, fpsbpts, cfg)
    losses, assigned = head_loss(cls, box, iou, gts_bottom, labels, cfg, assigned)
    return losses, dict(cls=cls, box=box, iou=iou, fpsbpts=fpsbpts, feats=x, assigned=assigned, encoder=enc, backbone=bb)
