"""Uni3DETRHead behind the reference's registry name, constructor and method signatures (ref:
projects/mmdet3d_plugin/models/dense_heads/uni3detr_head.py:311-918; upstream mmdet DETRHead).

`loss()` is restructured for the device: all decoder layers and scenes are matched in one u3d_match_cost + one u3d_lsa
launch, targets are built by index arithmetic (no boolean-mask indexing, no `.item()`), and the per-layer normalisers
are reduced across ranks in one all-reduce — values identical to the reference's per-layer / per-scene Python loops.
"""
import copy
import math

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from .. import native as nv
from ..registry import BBOX_ASSIGNERS, BBOX_CODERS, HEADS, LOSSES, TRANSFORMER
from .bbox import bbox_overlaps_3d_aligned, bbox_overlaps_nearest_3d, denormalize_bbox, normalize_bbox
from .transformer import colsum, inverse_sigmoid, run_sequential


def reduce_mean_(t):
    """In-place mean over ranks (mmdet `reduce_mean` for a whole vector at once); identity without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t.div_(dist.get_world_size())
        dist.all_reduce(t)
    return t


FUSED_BOX_DECODE = True
FUSED_DET_LOSS = True


class _DetLoss(torch.autograd.Function):
    """[L,4] per-layer (loss_cls, loss_bbox, loss_iou, loss_iou_pred) on the fused HIP kernels; gradients w.r.t. the class logits,
    box codes and IoU logits come from the in-kernel forward-mode derivatives (uni3detr_amd/csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, cls, box, iou_logit, tgt, lab, w, cls_avg, npos, code_w, alpha, w_cls, w_box, w_iou, eps):
        cls, box, iou_logit = cls.contiguous(), box.contiguous(), iou_logit.contiguous()
        tgt, lab, w = tgt.contiguous().float(), lab.contiguous().long(), w.contiguous().float()
        L, m, _ = cls.shape
        boxes = nv.denormalize_boxes(box.view(L * m, -1))
        iou_true = nv.iou3d_rotated_aligned(boxes, tgt.view(L * m, -1)[:, :7].contiguous())         # detached target, as the reference
        cls_avg, npos, code_w = cls_avg.contiguous(), npos.contiguous(), code_w.contiguous()
        out = nv.det_loss_fwd(cls, box, iou_logit, tgt, lab, w, iou_true, cls_avg, npos, code_w, alpha, w_cls, w_box, w_iou, eps)
        ctx.save_for_backward(cls, box, iou_logit, tgt, lab, w, iou_true, cls_avg, npos, code_w)
        ctx.cfg = (alpha, w_cls, w_box, w_iou, eps)
        return out

    @staticmethod
    def backward(ctx, gout):
        cls, box, iou_logit, tgt, lab, w, iou_true, cls_avg, npos, code_w = ctx.saved_tensors
        dcls, dbox, diou = nv.det_loss_bwd(cls, box, iou_logit, tgt, lab, w, iou_true, cls_avg, npos, code_w, gout.contiguous().float(),
                                           *ctx.cfg)
        return (dcls, dbox, diou) + (None,) * 11


FUSED_LOSS_TARGETS = True
FUSED_QUERY_EMBED = True


class _QueryEmbed(torch.autograd.Function):
    """Query assembly (ref: uni3detr_head.py:436-455) in one launch each way (u3d_query_embed_fwd / _bwd): the G groups' content
    embeddings and reference logits - group 0 = (tgt_embed[:nq], refpoint_embed), groups g >= 1 = (tgt_embed[nq:],
    inverse_sigmoid(sampled points)) - as query_embeds [B, G*nq, 259] AND as its two column blocks (what the transformer slices out
    of it again: the slices' backward would be two zero-fills and a full-tensor add)."""

    @staticmethod
    def forward(ctx, tgt, anchor, fps, rnd, groups):
        anchor = anchor.contiguous()
        qe, q, r, rs = nv.query_embed_fwd(tgt.contiguous(), anchor, fps.contiguous(), None if rnd is None else rnd.contiguous(), groups)
        ctx.dims = (fps.shape[0], anchor.shape[0], groups, tgt.shape[1])
        ctx.set_materialize_grads(False)          # unused outputs arrive as None, not as zero-filled tensors
        ctx.save_for_backward(anchor)             # rs = sigmoid(ref) is differentiable: the first layer's box decode reads the anchors through it
        return qe, q, r, rs

    @staticmethod
    def backward(ctx, dqe, dq, dr, drs):
        B, nq, G, c = ctx.dims
        (anchor,) = ctx.saved_tensors
        f = lambda t: None if t is None else t.contiguous().float()
        if dqe is None and dq is None and dr is None and drs is None:
            return None, None, None, None, None
        dev = next(t for t in (dqe, dq, dr, drs) if t is not None).device
        dt, da = nv.query_embed_bwd(f(dqe), f(dq), f(dr), f(drs), anchor, B, nq, G, c, dev)
        return dt, da, None, None, None


class _BoxDecode(torch.autograd.Function):
    """Regression-branch output + reference point -> normalised box code in ONE launch each way (u3d_box_decode_fwd/_bwd; ref:
    uni3detr_head.py:475-490).  tmp [..., code] f32|bf16, ref [..., 3] sigmoid space."""

    @staticmethod
    def forward(ctx, tmp, ref, pc_range):
        t2 = tmp.reshape(-1, tmp.shape[-1]).contiguous()
        r2 = ref.reshape(-1, 3).contiguous().float()
        ctx.save_for_backward(t2, r2)
        ctx.pc_range, ctx.shape, ctx.ref_shape, ctx.ref_dtype = pc_range, tmp.shape, ref.shape, ref.dtype
        return nv.box_decode_fwd(t2, r2, pc_range).view(tmp.shape)

    @staticmethod
    def backward(ctx, dout):
        t2, r2 = ctx.saved_tensors
        dtmp, dref = nv.box_decode_bwd(t2, r2, dout.reshape(t2.shape).contiguous().float(), ctx.pc_range, want_dref=ctx.needs_input_grad[1])
        return dtmp.view(ctx.shape), (None if dref is None else dref.view(ctx.ref_shape).to(ctx.ref_dtype)), None


def layer_sums(x):
    """x [L, ...] -> [L] sums of everything but the first dim, as a GEMV (graph-replay-safe, see transformer.colsum)."""
    return colsum(x.reshape(x.shape[0], -1).t())


def _clones(m, n):
    return nn.ModuleList(copy.deepcopy(m) for _ in range(n))


@HEADS.register_module()
class Uni3DETRHead(nn.Module):
    def __init__(self, num_classes, in_channels, num_query=100, num_reg_fcs=2, transformer=None, sync_cls_avg_factor=False,
                 positional_encoding=None, loss_cls=None, loss_bbox=dict(type="RotatedIoU3DLoss", loss_weight=1.0),
                 loss_iou=dict(type="RotatedIoU3DLoss", loss_weight=1.0), train_cfg=None, test_cfg=None, init_cfg=None,
                 with_box_refine=False, as_two_stage=False, bbox_coder=None, num_cls_fcs=2, code_weights=None,
                 post_processing=None, gt_repeattimes=1, code_size=10, **kwargs):
        super().__init__()
        if as_two_stage:
            raise NotImplementedError("as_two_stage is not used by any shipped Uni3DETR config")
        self.with_box_refine, self.as_two_stage = with_box_refine, as_two_stage
        self.code_size = code_size
        cw = code_weights if code_weights is not None else [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2]
        self.bbox_coder = BBOX_CODERS.build(bbox_coder)
        self.pc_range = self.bbox_coder.pc_range
        self.num_cls_fcs = num_cls_fcs - 1
        # --- what upstream DETRHead.__init__ sets up (no loss-weight == cost-weight assertion: the shipped configs
        #     use 1.5 vs 2.0)
        self.bg_cls_weight = 0
        self.sync_cls_avg_factor = sync_cls_avg_factor
        self.num_query, self.num_classes, self.in_channels = num_query, num_classes, in_channels
        self.num_reg_fcs, self.train_cfg, self.test_cfg = num_reg_fcs, train_cfg, test_cfg
        if train_cfg:
            self.assigner = BBOX_ASSIGNERS.build(train_cfg["assigner"])
        self.loss_cls = LOSSES.build(loss_cls)
        self.loss_bbox = LOSSES.build(loss_bbox)
        self.loss_iou = LOSSES.build(loss_iou)
        self.cls_out_channels = num_classes
        self.transformer = TRANSFORMER.build(transformer)
        self.embed_dims = self.transformer.embed_dims
        self._init_layers()
        self.code_weights = nn.Parameter(torch.tensor(cw, dtype=torch.float32), requires_grad=False)
        self.fp16_enabled = False
        self.post_processing = post_processing
        self.gt_repeattimes = gt_repeattimes

    def _init_layers(self):
        D = self.embed_dims
        cls = []
        for _ in range(self.num_reg_fcs):
            cls += [nn.Linear(D, D), nn.LayerNorm(D), nn.ReLU(inplace=True)]
        cls.append(nn.Linear(D, self.cls_out_channels))
        reg = []
        for _ in range(self.num_reg_fcs):
            reg += [nn.Linear(D, D), nn.ReLU()]
        reg.append(nn.Linear(D, self.code_size))
        iou = []
        for _ in range(self.num_reg_fcs):
            iou += [nn.Linear(D, D), nn.ReLU()]
        iou.append(nn.Linear(D, 1))
        n = self.transformer.decoder.num_layers
        mk = _clones if self.with_box_refine else (lambda m, k: nn.ModuleList([m for _ in range(k)]))
        self.cls_branches = mk(nn.Sequential(*cls), n)
        self.reg_branches = mk(nn.Sequential(*reg), n)
        self.iou_branches = mk(nn.Sequential(*iou), n)
        self.tgt_embed = nn.Embedding(self.num_query * 2, D)
        self.refpoint_embed = nn.Embedding(self.num_query, 3)

    def init_weights(self):
        """Never reached by the shipped training flow (Uni3DETR.init_weights is a bare return)."""
        self.transformer.init_weights()
        if self.loss_cls.use_sigmoid:
            b = float(-math.log((1 - 0.01) / 0.01))
            for m in self.cls_branches:
                nn.init.constant_(m[-1].bias, b)

    # ------------------------------------------------------------------------------------------
    def forward(self, pts_feats, img_metas, fpsbpts, rand_points=None):
        """pts_feats [B,C,D,H,W]; fpsbpts [B,2*nq,3] in [0,1].  Train layout (3 groups) iff pts_feats.requires_grad — the
        reference's switch (ref :443) — else 4 groups with a random-point group (`rand_points` lets tests inject it)."""
        nq = self.num_query
        tgt, anchor = self.tgt_embed.weight, self.refpoint_embed.weight
        B = fpsbpts.shape[0]
        if not pts_feats.requires_grad and rand_points is None:
            rand_points = torch.rand(fpsbpts.shape, device=fpsbpts.device)[:, :nq, :]
        if (FUSED_QUERY_EMBED and tgt.is_cuda and tgt.dtype == torch.float32 and anchor.dtype == torch.float32 and fpsbpts.dtype == torch.float32
                and tuple(fpsbpts.shape[1:]) == (2 * nq, 3) and tgt.shape[0] == 2 * nq and anchor.shape == (nq, 3)
                and (pts_feats.requires_grad or (rand_points.dtype == torch.float32 and tuple(rand_points.shape) == (B, nq, 3)))):
            query_embeds, q_part, r_part, r_sig = _QueryEmbed.apply(tgt, anchor, fpsbpts, None if pts_feats.requires_grad else rand_points,
                                                                    3 if pts_feats.requires_grad else 4)
            query_embeds._u3d_parts = (q_part, r_part, r_sig)   # Uni3DETRTransformer.forward takes these instead of slicing
        else:
            refs = [anchor.unsqueeze(0).expand(B, -1, -1), inverse_sigmoid(fpsbpts)]
            tgts = [tgt[:nq], tgt[nq:], tgt[nq:]]
            if not pts_feats.requires_grad:
                refs.append(inverse_sigmoid(rand_points))
                tgts.append(tgt[nq:])
            tgt_all = torch.cat(tgts)
            query_embeds = torch.cat([tgt_all.unsqueeze(0).expand(B, -1, -1), torch.cat(refs, 1).to(tgt_all.dtype)], -1)
        dec = self.transformer.decoder
        if self.with_box_refine and hasattr(dec, "forward_bf"):
            object.__setattr__(dec, "_head_branches", (self.cls_branches, self.iou_branches))     # the fused bf16 path runs them per layer
            object.__setattr__(dec, "_pc_range", tuple(float(v) for v in self.pc_range) if FUSED_BOX_DECODE else None)      # ... and the box decode
        hs, init_reference, inter_references = self.transformer(
            pts_feats, query_embeds, nq, reg_branches=self.reg_branches if self.with_box_refine else None, img_metas=img_metas)
        hs = hs.permute(0, 2, 1, 3)                                                   # [L,B,N,C]
        pr = self.pc_range
        classes, coords, ious = [], [], []
        for lvl in range(hs.shape[0]):
            ref_s = init_reference if lvl == 0 else inter_references[lvl - 1]
            h = hs[lvl]
            sc = getattr(self.transformer.decoder, "_states_c", None)
            if sc is not None and len(sc) == hs.shape[0] and torch.is_autocast_enabled():
                h = sc[lvl]                    # the decoder's own compute-dtype copy of this state (one cast serves every branch)
            reg = getattr(self.transformer.decoder, "_reg_outputs", None)
            if self.with_box_refine and reg is not None and len(reg) == hs.shape[0]:
                tmp = reg[lvl]                 # the decoder already ran reg_branches[lvl] on this very state to refine its points
            else:
                tmp = run_sequential(self.reg_branches[lvl], h)
            assert ref_s.shape[-1] == 3
            co = getattr(dec, "_coord_outputs", None)
            if co is not None and len(co) == hs.shape[0] and self.with_box_refine:
                coords.append(co[lvl])         # decoded by the decoder layer's own tail launch (fused_decoder._RefineDecode)
            elif FUSED_BOX_DECODE and tmp.is_cuda and tmp.dtype in (torch.float32, torch.bfloat16) and tmp.shape[-1] <= 16:
                coords.append(_BoxDecode.apply(tmp, ref_s, tuple(float(v) for v in pr)))
            else:
                reference = inverse_sigmoid(ref_s)
                t = tmp.float().unbind(-1)
                rf = reference.unbind(-1)
                x_ = (t[0] + rf[0]).sigmoid() * (pr[3] - pr[0]) + pr[0]
                y_ = (t[1] + rf[1]).sigmoid() * (pr[4] - pr[1]) + pr[1]
                z_ = (t[4] + rf[2]).sigmoid() * (pr[5] - pr[2]) + pr[2]
                coords.append(torch.stack([x_, y_, t[2], t[3], z_, *t[5:]], -1))
            cls_o, iou_o = getattr(dec, "_cls_outputs", None), getattr(dec, "_iou_outputs", None)
            if cls_o is not None and iou_o is not None and len(cls_o) == hs.shape[0]:
                classes.append(cls_o[lvl])          # produced by the fused decoder layer on this very state
                ious.append(iou_o[lvl])
            else:
                classes.append(run_sequential(self.cls_branches[lvl], h).float())
                ious.append(run_sequential(self.iou_branches[lvl], h).float())
        return {"all_cls_scores": torch.stack(classes), "all_bbox_preds": torch.stack(coords), "all_iou_preds": torch.stack(ious)}

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _bbox_to_loss(b):
        return torch.stack((b[..., 0] - b[..., 3] / 2, b[..., 1] - b[..., 4] / 2, b[..., 2] - b[..., 5] / 2,
                            b[..., 0] + b[..., 3] / 2, b[..., 1] + b[..., 4] / 2, b[..., 2] + b[..., 5] / 2), dim=-1)

    @property
    def gt_dim(self):
        """Columns of a regression target: 7 (x, y, z, dx, dy, dz, yaw), or 9 with the velocity pair when the box code has 10 entries
        (nuScenes).  The reference slices its targets to 7 columns (`zeros_like(bbox_pred)[..., :7]`, uni3detr_head.py:557; the
        upstream `[..., :9]` line sits commented out right above it) and then feeds them to a 10-column L1 loss and a 10-entry
        code_weights vector (uni3detr_head.py:684-687) - with code_size 10 that is a shape error for 7- and 9-column GT alike, i.e.
        uni3detr_nuscenes.py does not train as shipped.  The target here is as wide as the code asks (the upstream DETR3D semantics the
        file derives from); 7-column GT gets zero velocities."""
        return 9 if int(self.code_size) >= 10 else 7

    def _pack_gts(self, gt_bboxes_list, gt_labels_list, device):
        if isinstance(gt_bboxes_list, dict):       # pre-packed static buffers: dict(gt [cap,gt_dim] gravity-centre, labels int32 [cap], gt_off int32 [B+1], gmax int)
            d = gt_bboxes_list
            return d["gt"], d["labels"], d["gt_off"], int(d["gmax"])
        gd = self.gt_dim
        gts = []
        for g in gt_bboxes_list:
            if hasattr(g, "gravity_center"):
                g = torch.cat((g.gravity_center, g.tensor[:, 3:]), dim=1)
            g = g.to(device=device, dtype=torch.float32)[:, :gd]
            if g.shape[1] < gd:
                g = torch.nn.functional.pad(g, (0, gd - g.shape[1]))
            gts.append(g)
        lens = [int(g.shape[0]) for g in gts]
        off = [0]
        for n in lens:
            off.append(off[-1] + n)
        gt = torch.cat(gts) if sum(lens) else torch.zeros((0, gd), device=device)
        labels = torch.cat([l.to(device) for l in gt_labels_list]).int() if sum(lens) else torch.zeros((0,), dtype=torch.int32, device=device)
        return gt.contiguous(), labels.contiguous(), torch.tensor(off, dtype=torch.int32, device=device), max(lens) if lens else 0

    def pack_gts(self, gt_bboxes_list, gt_labels_list, device):
        """Pack a batch's GTs once (static input buffers for hipGraph replay); pass the result as `gt_bboxes_3d`."""
        gt, labels, gt_off, gmax = self._pack_gts(gt_bboxes_list, gt_labels_list, device)
        return dict(gt=gt, labels=labels, gt_off=gt_off, gmax=gmax)

    def loss_targets(self, gt_bboxes_list, gt_labels_list, preds_dicts):
        """Stage 1 of loss(): matching + target construction for all layers/scenes (no cross-rank communication).
        Returns a dict with the assignment, targets and the LOCAL per-layer positive counts."""
        cls_all = preds_dicts["all_cls_scores"].float()
        box_all = preds_dicts["all_bbox_preds"].float()
        L, B, Q, C = cls_all.shape
        dev = cls_all.device
        gt, labels, gt_off, gmax = self._pack_gts(gt_bboxes_list, gt_labels_list, dev)
        gt7 = gt if gt.shape[1] == 7 else gt[:, :7].contiguous()          # matching sees the 7 geometric columns (ref: match_cost.py:19-30, 91-97)
        asg = self.assigner.assign_batched(cls_all, box_all, gt7, labels, gt_off, gmax, self.num_query)          # int32 [L,B,Q]
        if (FUSED_LOSS_TARGETS and asg.is_cuda and gt.shape[0] and gt.dtype == torch.float32 and gt.is_contiguous()
                and labels.dtype == torch.int32 and gt_off.dtype == torch.int32):
            a64, w, tgt, lab, npos = nv.loss_targets(asg.contiguous(), gt, labels.contiguous(), gt_off, C)       # one launch (u3d_loss_targets)
            return dict(asg=a64, w=w, tgt=tgt, lab=lab, num_pos=npos)
        asg = asg.long()
        pos = asg > 0
        w = pos.to(torch.float32)
        if gt.shape[0]:
            gidx = (gt_off[:-1].long().view(1, B, 1) + (asg - 1).clamp(min=0))
            tgt = gt[gidx] * w.unsqueeze(-1)                                           # zeros for background rows
            lab = torch.where(pos, labels.long()[gidx], torch.full_like(asg, C))
        else:
            tgt = box_all.new_zeros((L, B, Q, gt.shape[1]))
            lab = torch.full_like(asg, C)
        return dict(asg=asg, w=w, tgt=tgt, lab=lab, num_pos=layer_sums(w))

    def loss_from_targets(self, preds_dicts, T, num_pos):
        """Stage 2 of loss(): the 4 losses x L layers given targets and the (rank-averaged) positive counts [L]."""
        cls_all = preds_dicts["all_cls_scores"].float()
        box_all = preds_dicts["all_bbox_preds"].float()
        iou_all = preds_dicts["all_iou_preds"].float()
        L, B, Q, C = cls_all.shape
        w, tgt, lab = T["w"], T["tgt"], T["lab"]
        npos = num_pos.clamp(min=1)
        cls_avg = npos if self.sync_cls_avg_factor else T["num_pos"].clamp(min=1)      # (one launch, not two, in the usual synced case)
        from .losses import IoU3DLoss, L1Loss, SoftFocalLoss, _EPS32
        if (FUSED_DET_LOSS and cls_all.is_cuda and type(self.loss_cls) is SoftFocalLoss and type(self.loss_bbox) is L1Loss
                and type(self.loss_iou) is IoU3DLoss and self.loss_cls.reduction == self.loss_bbox.reduction == self.loss_iou.reduction == "mean"
                and float(self.loss_cls.gamma) == 2.0 and box_all.shape[-1] in (8, 10) and tgt.shape[-1] == box_all.shape[-1] - 1
                and self.code_weights.numel() == box_all.shape[-1]):
            # all four losses of all layers: one HIP launch forward (+ a 12-value reduce), one backward (u3d_det_loss_fwd / _bwd)
            per = _DetLoss.apply(cls_all.reshape(L, B * Q, C), box_all.reshape(L, B * Q, -1), iou_all.reshape(L, B * Q),
                                 tgt.reshape(L, B * Q, -1), lab.reshape(L, B * Q), w.reshape(L, B * Q), cls_avg.float(), npos.float(),
                                 self.code_weights.detach().float(), float(self.loss_cls.alpha), float(self.loss_cls.loss_weight),
                                 float(self.loss_bbox.loss_weight), float(self.loss_iou.loss_weight), float(_EPS32))   # [L, 4]
            out = {"loss_cls": per[L - 1, 0], "loss_bbox": per[L - 1, 1], "loss_iou": per[L - 1, 2], "loss_iou_pred": per[L - 1, 3]}
            for i in range(L - 1):
                out[f"d{i}.loss_cls"], out[f"d{i}.loss_bbox"] = per[i, 0], per[i, 1]
                out[f"d{i}.loss_iou"], out[f"d{i}.loss_iou_pred"] = per[i, 2], per[i, 3]
            self._last_assigned = T["asg"]
            # sum of the 12 scalars as ONE reduction of `per`: a training step that back-propagates this instead of the python sum of
            # the dict values skips 12 select-backward (zero-fill + copy) and 11 accumulate launches
            self._loss_total = per.sum()
            return out
        self._loss_total = None
        ntgt = normalize_bbox(tgt, self.pc_range)
        b3d = denormalize_bbox(box_all, self.pc_range)
        iou_bev = bbox_overlaps_nearest_3d(b3d, tgt, is_aligned=True)                  # [L,B,Q]
        pc, tc = b3d.unbind(-1), tgt.unbind(-1)                                        # z extents as _bbox_to_loss builds them (:671-674)
        z1, z2, z3, z4 = pc[2] - pc[5] / 2, pc[2] + pc[5] / 2, tc[2] - tc[5] / 2, tc[2] + tc[5] / 2
        iou_z = (torch.min(z2, z4) - torch.max(z1, z3)).clamp(min=0) / (torch.max(z2, z4) - torch.min(z1, z3))
        quality = (iou_bev + iou_z) / 2                                                # not detached
        bw = w.unsqueeze(-1) * self.code_weights
        iou_true = bbox_overlaps_3d_aligned(b3d, tgt).view(L, B, Q)
        from .losses import IoU3DLoss, L1Loss, SoftFocalLoss, _EPS32
        if (type(self.loss_cls) is SoftFocalLoss and type(self.loss_bbox) is L1Loss and type(self.loss_iou) is IoU3DLoss
                and self.loss_cls.reduction == self.loss_bbox.reduction == self.loss_iou.reduction == "mean"):
            # all layers at once: the same arithmetic as the per-layer module calls below (weights of 1 on every label row,
            # `mean` reduction with avg_factor), written on [L,B,Q,*] tensors — a third of the launches
            a, gmm = self.loss_cls.alpha, self.loss_cls.gamma
            ps = cls_all.sigmoid()
            soft = F.one_hot(lab, C + 1)[..., :C].to(cls_all.dtype) * quality.unsqueeze(-1)
            fw = ((1 - a) + (2 * a - 1) * soft) * (soft - ps).pow(gmm)
            l_cls = layer_sums(F.binary_cross_entropy_with_logits(cls_all, soft, reduction="none") * fw) / (cls_avg + _EPS32) * self.loss_cls.loss_weight
            l_box = layer_sums((box_all[..., :10] - ntgt[..., :10]).abs() * bw[..., :10]) / (npos + _EPS32) * self.loss_bbox.loss_weight
            l_iou = layer_sums((1 - iou_bev) * bw[..., :10].mean(-1)) / (npos + _EPS32) * self.loss_iou.loss_weight
            l_iou = l_iou + layer_sums((1 - iou_z) * bw[..., 0]) / npos
            l_ioup = layer_sums(F.binary_cross_entropy_with_logits(iou_all.squeeze(-1), iou_true, reduction="none") * bw[..., 0]) / npos * 1.2
            out = {"loss_cls": l_cls[-1], "loss_bbox": l_box[-1], "loss_iou": l_iou[-1], "loss_iou_pred": l_ioup[-1]}
            for i in range(L - 1):
                out[f"d{i}.loss_cls"], out[f"d{i}.loss_bbox"] = l_cls[i], l_box[i]
                out[f"d{i}.loss_iou"], out[f"d{i}.loss_iou_pred"] = l_iou[i], l_ioup[i]
            self._last_assigned = T["asg"]
            return out
        losses_cls, losses_bbox, losses_iou, losses_ioup = [], [], [], []
        for l in range(L):
            lc = self.loss_cls(cls_all[l].reshape(-1, C), [lab[l].reshape(-1), quality[l].reshape(-1)], w.new_ones(B * Q), avg_factor=cls_avg[l])
            lb = self.loss_bbox(box_all[l].reshape(B * Q, -1)[:, :10], ntgt[l].reshape(B * Q, -1)[:, :10], bw[l].reshape(B * Q, -1)[:, :10], avg_factor=npos[l])
            li = self.loss_iou(b3d[l].reshape(B * Q, -1)[:, :10], tgt[l].reshape(B * Q, -1)[:, :10], bw[l].reshape(B * Q, -1)[:, :10], avg_factor=npos[l])
            li = li + torch.sum((1 - iou_z[l]) * bw[l][..., 0]) / npos[l]
            lp = torch.sum(F.binary_cross_entropy_with_logits(iou_all[l].reshape(-1), iou_true[l].reshape(-1), reduction="none")
                           * bw[l][..., 0].reshape(-1)) / npos[l] * 1.2
            losses_cls.append(lc); losses_bbox.append(lb); losses_iou.append(li); losses_ioup.append(lp)
        out = {"loss_cls": losses_cls[-1], "loss_bbox": losses_bbox[-1], "loss_iou": losses_iou[-1], "loss_iou_pred": losses_ioup[-1]}
        for i in range(L - 1):
            out[f"d{i}.loss_cls"], out[f"d{i}.loss_bbox"] = losses_cls[i], losses_bbox[i]
            out[f"d{i}.loss_iou"], out[f"d{i}.loss_iou_pred"] = losses_iou[i], losses_ioup[i]
        self._last_assigned = T["asg"]
        return out

    def loss(self, gt_bboxes_list, gt_labels_list, preds_dicts, gt_bboxes_ignore=None):
        assert gt_bboxes_ignore is None, f"{self.__class__.__name__} only supports for gt_bboxes_ignore setting to None."
        T = self.loss_targets(gt_bboxes_list, gt_labels_list, preds_dicts)
        num_pos = reduce_mean_(T["num_pos"].clone())           # [L]; one message for all layers (ref: 2 scalar all-reduces per layer)
        return self.loss_from_targets(preds_dicts, T, num_pos)

    def soft_nms(self, boxes, scores, gaussian_sigma=0.3, prune_threshold=1e-3):
        """Gaussian soft-NMS with rotated 3-D IoU over ONE set of boxes (ref :796-823) on the device kernel (u3d_soft_nms)."""
        idx, sc, _ = nv.soft_nms_classwise(boxes, scores, torch.zeros(boxes.shape[0], dtype=torch.int32, device=boxes.device), 1,
                                           gaussian_sigma, prune_threshold)
        return idx, sc

    def get_bboxes(self, preds_dicts, img_metas, rescale=False):
        """Decode + post-processing on the device (ref :827-918).  Returns [[boxes [n,7] bottom-centre, scores, labels], ...]."""
        from .. import native as nv
        preds = self.bbox_coder.decode(preds_dicts)
        pp = self.post_processing
        ret = []
        for p in preds:
            boxes = p["bboxes"].clone()
            boxes[:, 2] = boxes[:, 2] - boxes[:, 5] * 0.5          # gravity centre -> bottom centre (ref :842)
            scores, labels = p["scores"], p["labels"]
            if pp is not None:
                if pp["type"] == "nms":
                    keep = nv.nms3d_classwise(boxes, scores, labels, pp["nms_thr"])
                    boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
                elif pp["type"] == "soft_nms":
                    # all classes in one launch (one workgroup per class); result class-major like the reference's per-class loop
                    ki, scores, labels = nv.soft_nms_classwise(boxes, scores, labels, self.num_classes, pp["gaussian_sigma"], pp["prune_threshold"])
                    boxes = boxes[ki]
                elif pp["type"] == "box_merging":
                    # ref :881-891: nms_boxes_3d_merge_only(..., overlapped_thres=0.1, top_k=-1): score sort, greedy same-class merge,
                    # kept boxes replaced by the median of what they absorbed - on the device (u3d_box_merge)
                    order = torch.argsort(-scores, stable=True)
                    boxes, scores, labels = boxes[order], scores[order], labels[order]
                    merged, keep = nv.box_merge(boxes[:, :7], labels, 0.1)
                    boxes = torch.cat([merged, boxes[:, 7:]], 1)[keep] if boxes.shape[1] > 7 else merged[keep]
                    scores, labels = scores[keep], labels[keep]
                else:
                    raise NotImplementedError(pp["type"] + " not implemented.")
                if "score_thr" in pp:
                    thr = pp["score_thr"]
                    if isinstance(thr, (list, tuple)):
                        assert len(thr) == self.num_classes
                        ind = scores > scores.new_tensor(list(thr))[labels]
                    else:
                        ind = scores > thr
                    boxes, scores, labels = boxes[ind], scores[ind], labels[ind]
                if "num_thr" in pp:
                    ind = torch.argsort(-scores)[: pp["num_thr"]]
                    boxes, scores, labels = boxes[ind], scores[ind], labels[ind]
            ret.append([boxes, scores, labels])
        return ret
