"""SoftFocalLoss / IoU3DLoss / L1Loss behind the reference's registry names (ref:
projects/mmdet3d_plugin/models/losses/rdiouloss.py:93-223; upstream mmdet L1Loss / weight_reduce_loss)."""
import torch
import torch.nn.functional as F
from torch import nn

from ..registry import LOSSES
from .bbox import bbox_overlaps_nearest_3d

_EPS32 = torch.finfo(torch.float32).eps


def weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == "mean" else (loss.sum() if reduction == "sum" else loss)
    if reduction == "mean":
        return loss.sum() / (avg_factor + _EPS32)
    if reduction == "none":
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def soft_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction="mean", avg_factor=None):
    """Quality-focal classification loss; target = (labels, soft score); background label == num_classes."""
    labels, score = target
    p = pred.sigmoid()
    onehot = F.one_hot(labels, pred.shape[1] + 1)[:, : pred.shape[1]].to(pred.dtype)
    soft = onehot * score[:, None]
    pt = soft - p
    fw = ((1 - alpha) + (2 * alpha - 1) * soft) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, soft, reduction="none") * fw
    return weight_reduce_loss(loss, None if weight is None else weight.view(-1, 1), reduction, avg_factor)


@LOSSES.register_module()
class SoftFocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, "Only sigmoid focal loss supported now."
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, "none", "mean", "sum")
        return self.loss_weight * soft_focal_loss(pred, target, weight, self.gamma, self.alpha, reduction_override or self.reduction, avg_factor)


@LOSSES.register_module()
class IoU3DLoss(nn.Module):
    """1 - nearest-BEV IoU (axis-snapped, height ignored) despite the name."""

    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        # the reference's `not any(weight > 0)` early-out returns pred.sum()*weight.sum() == 0 with zero gradients — the
        # same value and gradients the general formula gives, so no host-synchronising branch is taken here.
        assert reduction_override in (None, "none", "mean", "sum")
        if weight is not None and weight.dim() > 1:
            weight = weight.mean(-1)
        loss = 1 - bbox_overlaps_nearest_3d(pred, target, is_aligned=True)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction_override or self.reduction, avg_factor)


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        return self.loss_weight * weight_reduce_loss((pred - target).abs(), weight, reduction_override or self.reduction, avg_factor)
