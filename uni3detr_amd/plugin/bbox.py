"""Box codes, IoU helpers, match costs, assigner, coder (ref: projects/mmdet3d_plugin/core/bbox/util.py:8-80,
core/bbox/match_costs/match_cost.py:9-31,85-97, core/bbox/assigners/hungarian_assigner_3d.py:17-151,
core/bbox/coders/nms_free_coder.py:8-136; upstream mmdet3d nearest_bev / bbox_overlaps_nearest_3d)."""
import math

import torch

from .. import native as nv
from ..registry import BBOX_ASSIGNERS, BBOX_CODERS, MATCH_COST


def normalize_bbox(bboxes, pc_range=None):
    """(cx,cy,cz,dx,dy,dz,yaw[,vx,vy]) -> code (cx,cy,log dy,log dx,cz,log dz,sin r,cos r[,vx,vy]), r = -yaw - pi/2.
    unbind/stack instead of eight slices + cat: one backward node instead of eight zero-fill + copy pairs."""
    c = bboxes.unbind(-1)
    rot = -c[6] - math.pi / 2
    sizes = (torch.stack((c[4], c[3], c[5]), -1) + 1e-5).log().unbind(-1)
    parts = [c[0], c[1], sizes[0], sizes[1], c[2], sizes[2], rot.sin(), rot.cos()]
    if bboxes.size(-1) > 7:
        parts += [c[7], c[8]]
    return torch.stack(parts, dim=-1)


def denormalize_bbox(codes, pc_range=None, version=0.8):
    c = codes.unbind(-1)
    rot = -torch.atan2(c[6], c[7]) - math.pi / 2
    sizes = torch.stack((c[3], c[2], c[5]), -1).exp().unbind(-1)
    parts = [c[0], c[1], c[4], sizes[0], sizes[1], sizes[2], rot]
    if codes.size(-1) > 8:
        parts += [c[8], c[9]]
    return torch.stack(parts, dim=-1)


def nearest_bev(boxes):
    c = boxes.unbind(-1)
    yaw = c[6]
    r = (yaw - torch.floor(yaw / math.pi + 0.5) * math.pi).abs()
    swap = r > math.pi / 4
    w = torch.where(swap, c[4], c[3])
    h = torch.where(swap, c[3], c[4])
    return torch.stack((c[0] - w / 2, c[1] - h / 2, c[0] + w / 2, c[1] + h / 2), dim=-1)


def _iou_xyxy(a, b, aligned, eps=1e-6):
    if not aligned:
        a, b = a[..., :, None, :], b[..., None, :, :]
    ax1, ay1, ax2, ay2 = a.unbind(-1)
    bx1, by1, bx2, by2 = b.unbind(-1)
    ov = (torch.min(ax2, bx2) - torch.max(ax1, bx1)).clamp(min=0) * (torch.min(ay2, by2) - torch.max(ay1, by1)).clamp(min=0)
    union = ((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - ov).clamp(min=eps)
    return ov / union


def bbox_overlaps_nearest_3d(b1, b2, mode="iou", is_aligned=False, coordinate="lidar"):
    """Axis-snapped BEV IoU (height ignored) — what `IoU3DCost` / `IoU3DLoss` really compute."""
    return _iou_xyxy(nearest_bev(b1[..., :7]), nearest_bev(b2[..., :7]), is_aligned)


class _RotIoU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return nv.iou3d_rotated_aligned(a.detach().float().contiguous(), b.detach().float().contiguous())

    @staticmethod
    def backward(ctx, g):
        return None, None


def bbox_overlaps_3d_aligned(b1, b2):
    """diag(bbox_overlaps_3d(b1, b2, coordinate='lidar')) on the HIP kernel; no gradient (the reference detaches it)."""
    return _RotIoU.apply(b1[..., :7].reshape(-1, 7), b2[..., :7].reshape(-1, 7))


@MATCH_COST.register_module()
class FocalLossCost:
    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12, binary_input=False):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MATCH_COST.register_module()
class BBox3DL1Cost:
    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        return torch.cdist(bbox_pred, gt_bboxes, p=1) * self.weight


@MATCH_COST.register_module()
class IoU3DCost:
    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        return (1 - bbox_overlaps_nearest_3d(bbox_pred, gt_bboxes)) * self.weight


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


@BBOX_ASSIGNERS.register_module()
class HungarianAssigner3D:
    """cost = focal + L1 + (1 - nearest-BEV IoU), then one exact assignment per 300-query group — all on the device
    (u3d_match_cost + u3d_lsa) instead of the reference's `.cpu()` + scipy round trip (ref :110-142)."""

    def __init__(self, cls_cost=dict(type="ClassificationCost", weight=1.0), reg_cost=dict(type="BBoxL1Cost", weight=1.0),
                 iou_cost=dict(type="IoUCost", weight=0.0), pc_range=None):
        self.cls_cost, self.reg_cost, self.iou_cost = MATCH_COST.build(cls_cost), MATCH_COST.build(reg_cost), MATCH_COST.build(iou_cost)
        for c, t in ((self.cls_cost, FocalLossCost), (self.reg_cost, BBox3DL1Cost), (self.iou_cost, IoU3DCost)):
            if not isinstance(c, t):
                raise NotImplementedError(f"the HIP matcher implements {t.__name__}; got {type(c).__name__}")
        self.pc_range = pc_range

    def assign_batched(self, cls_all, box_all, gt, labels, gt_off, gmax, num_query):
        """cls_all [L,B,Q,C], box_all [L,B,Q,code] (f32), gt [sumG,7] gravity-centre, labels int32 [sumG], gt_off int32 [B+1]
        (device), gmax = max GTs per scene (host int) -> assigned int32 [L,B,Q] (0 = background, else 1-based GT)."""
        L, B, Q, _ = cls_all.shape
        if gmax == 0:
            return torch.zeros((L, B, Q), dtype=torch.int32, device=cls_all.device)
        cost = nv.match_cost(cls_all.detach().float().contiguous(), box_all.detach().float().contiguous(), gt.float().contiguous(),
                             labels, gt_off, gmax, self.cls_cost.weight, self.reg_cost.weight, self.iou_cost.weight,
                             self.cls_cost.alpha, float(self.cls_cost.gamma))
        return nv.lsa(cost, gt_off, L, B, Q, num_query, gmax)

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, num_query, gt_bboxes_ignore=None, eps=1e-7, gt_repeattimes=1):
        """Single-scene drop-in (ref :53-151).  `gt_repeattimes` is accepted but never takes effect upstream either: the
        head passes it positionally into `eps`."""
        assert gt_bboxes_ignore is None, "Only case when gt_bboxes_ignore is None is supported."
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            inds = bbox_pred.new_full((num_bboxes,), 0 if num_gts == 0 else -1, dtype=torch.long)
            return AssignResult(num_gts, inds, None, labels=labels)
        gt_off = torch.tensor([0, num_gts], dtype=torch.int32, device=bbox_pred.device)
        a = self.assign_batched(cls_pred[None, None], bbox_pred[None, None], gt_bboxes, gt_labels.int(), gt_off, num_gts, num_query)
        inds = a[0, 0].long()
        pos = inds > 0
        labels[pos] = gt_labels[inds[pos] - 1]
        return AssignResult(num_gts, inds, None, labels=labels)


@BBOX_CODERS.register_module()
class NMSFreeCoder:
    """Test-time decode (ref: nms_free_coder.py:22-136): mean of decoder layers [1:], sigmoid, top-k over query x class,
    score^alpha * iou^(1-alpha), centre-range mask."""

    def __init__(self, pc_range, voxel_size=None, post_center_range=None, max_num=100, score_threshold=None, alpha=0.5, num_classes=10):
        self.pc_range, self.voxel_size, self.post_center_range = pc_range, voxel_size, post_center_range
        self.max_num, self.score_threshold, self.alpha, self.num_classes = max_num, score_threshold, alpha, num_classes

    def encode(self):
        pass

    def decode_single(self, cls_scores, bbox_preds, iou_preds):
        if self.post_center_range is None:
            raise NotImplementedError("only post_center_range is not None is supported (as in the reference)")
        cls_scores = cls_scores.sigmoid()
        # top-k with a PINNED tie order: descending score, equal scores by ascending (query, class) index - a stable descending sort keeps
        # equal elements in input order by contract on every device (the upstream `topk` leaves ties to the backend: ref :70)
        # Cost: a top-k for the cut, then the stable sort over the SELECTED elements only (everything >= the k-th score: k of them unless
        # the k-th score is tied) - not over all num_query x num_classes scores (nuScenes: 27 000 per scene)
        flat = cls_scores.reshape(-1)
        k = min(int(self.max_num), flat.numel())
        kth = flat.topk(k).values[-1]
        cand = (flat >= kth).nonzero().view(-1)                       # ascending (query, class) index
        scores, order = flat[cand].sort(descending=True, stable=True)
        scores, idx = scores[:k], cand[order][:k]
        labels = idx % self.num_classes
        bidx = torch.div(idx, self.num_classes, rounding_mode="floor")
        boxes = denormalize_bbox(bbox_preds[bidx], self.pc_range)
        ious = iou_preds.sigmoid()[bidx].reshape(-1)
        r = scores.new_tensor(self.post_center_range)
        keep = (boxes[..., :3] >= r[:3]).all(1) & (boxes[..., :3] <= r[3:]).all(1)
        if self.score_threshold:
            keep &= scores > self.score_threshold
        scores, ious = scores[keep], ious[keep]
        return dict(bboxes=boxes[keep], scores=scores ** self.alpha * ious ** (1 - self.alpha), labels=labels[keep], ious=ious)

    def decode(self, preds_dicts):
        cls = preds_dicts["all_cls_scores"][1:].mean(0)
        box = preds_dicts["all_bbox_preds"][1:].mean(0)
        iou = preds_dicts["all_iou_preds"][1:].mean(0)
        return [self.decode_single(cls[i], box[i], iou[i]) for i in range(cls.size(0))]
