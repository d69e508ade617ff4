"""Registry names of the reference plugin that NO shipped config selects, kept so that a user config naming them still builds
(SURVEY.md 8b-1): `RDIoULoss` (ref: models/losses/rdiouloss.py:13-91), `RotatedIoU3DCost`, `AxisAlignedIoU3DCost`, `RDIoUCost`,
`SoftFocalLossCost` (ref: core/bbox/match_costs/match_cost.py:34-128) and `get_rdiou` (ref: core/bbox/util.py:104-153).
Plain torch formulations: none of them is on the hot path.  Pinned by tests/golden/extra_costs.npz (generated from the reference's
own files) where the reference implements the arithmetic itself; the two costs that lean on mmcv / mmdet3d operators are checked
against closed forms."""
import torch
from torch import nn

from ..registry import LOSSES, MATCH_COST
from .losses import weight_reduce_loss


def get_rdiou(b1, b2):
    """Rotation-decoupled IoU of boxes (x, y, z, log l, log w, log h, yaw) that broadcast against each other on the leading dims:
    the IoU and the normalised centre distance of 4-D axis-aligned boxes over (x, 2y, 2z, t), where the rotation enters as the
    decoupled coordinate t1 = sin(a1) cos(a2), t2 = cos(a1) sin(a2) with unit extent.  Returns (u, rdiou)."""
    a1, a2 = b1[..., 6], b2[..., 6]
    ones = torch.ones_like(b1[..., 0] + b2[..., 0])
    scale = b1.new_tensor([1.0, 2.0, 2.0])
    c1 = torch.cat([(b1[..., :3] * scale).expand(*ones.shape, 3), (a1.sin() * a2.cos()).unsqueeze(-1)], -1)
    c2 = torch.cat([(b2[..., :3] * scale).expand(*ones.shape, 3), (a1.cos() * a2.sin()).unsqueeze(-1)], -1)
    e1 = torch.cat([b1[..., 3:6].exp().clamp(max=10).expand(*ones.shape, 3), ones.unsqueeze(-1)], -1)     # only the first box is clamped
    e2 = torch.cat([b2[..., 3:6].exp().expand(*ones.shape, 3), ones.unsqueeze(-1)], -1)
    lo1, hi1, lo2, hi2 = c1 - e1 / 2, c1 + e1 / 2, c2 - e2 / 2, c2 + e2 / 2
    inter = (torch.min(hi1, hi2) - torch.max(lo1, lo2)).clamp(min=0).prod(-1)
    hull = (torch.max(hi1, hi2) - torch.min(lo1, lo2)).clamp(min=0)
    union = e1.prod(-1) + e2.prod(-1) - inter
    return ((c2 - c1) ** 2).sum(-1) / (hull ** 2).sum(-1), inter / union


def _rdiou_term(b1, b2):
    u, r = get_rdiou(b1, b2)
    return 1 - (r - u).clamp(min=-1.0, max=1.0)


@LOSSES.register_module()
class RDIoULoss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, "none", "mean", "sum")
        if weight is not None and weight.dim() > 1:
            weight = weight.mean(-1)
        # (the reference's all-zero-weight early-out returns exactly what the general formula returns: zero with zero gradients)
        return self.loss_weight * weight_reduce_loss(_rdiou_term(pred, target), weight, reduction_override or self.reduction, avg_factor)


@MATCH_COST.register_module()
class RDIoUCost:
    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        return _rdiou_term(bbox_pred.unsqueeze(1), gt_bboxes.unsqueeze(0)) * self.weight


@MATCH_COST.register_module()
class SoftFocalLossCost:
    """Focal classification cost with the class probability tempered by iou^0.001 (so a zero-IoU pair costs like a miss)."""

    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12, binary_input=False):
        self.weight, self.alpha, self.gamma, self.eps, self.binary_input = weight, alpha, gamma, eps, binary_input

    def __call__(self, cls_pred, gt_labels, iou3d):
        q = cls_pred.sigmoid() * iou3d.pow(0.001)
        neg = -(1 - q + self.eps).log() * (1 - self.alpha) * q.pow(self.gamma)
        pos = -(q + self.eps).log() * self.alpha * (1 - q).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MATCH_COST.register_module()
class AxisAlignedIoU3DCost:
    """-IoU of axis-aligned boxes (x1, y1, z1, x2, y2, z2), pairwise [num_query, num_gt] (upstream AxisAlignedBboxOverlaps3D)."""

    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        p, g = bbox_pred.unsqueeze(1), gt_bboxes.unsqueeze(0)
        inter = (torch.min(p[..., 3:6], g[..., 3:6]) - torch.max(p[..., :3], g[..., :3])).clamp(min=0).prod(-1)
        vol = lambda b: (b[..., 3:6] - b[..., :3]).prod(-1)
        return -(inter / (vol(p) + vol(g) - inter).clamp(min=1e-6)) * self.weight


@MATCH_COST.register_module()
class RotatedIoU3DCost:
    """Rotated 3-D IoU of every (prediction, GT) pair times the weight - the IoU itself, not 1 - IoU, as the reference has it
    (upstream mmcv diff_iou_rotated_3d; here the device rotated-IoU kernel, costs carry no gradient)."""

    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        from .bbox import bbox_overlaps_3d_aligned
        m, n = bbox_pred.shape[0], gt_bboxes.shape[0]
        p = bbox_pred[:, None, :7].expand(m, n, 7).reshape(-1, 7)
        g = gt_bboxes[None, :, :7].expand(m, n, 7).reshape(-1, 7)
        return bbox_overlaps_3d_aligned(p.contiguous(), g.contiguous()).view(m, n) * self.weight
