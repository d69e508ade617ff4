"""ORACLE (test infrastructure).  CPU restatement of the mmdet / mmdet3d box helpers the hot path calls.

None of these live under /root/reference (they are mmdet 2.x / mmdet3d v1.0.0rc5 code, not installed here):
PARITY UNPINNED against upstream binaries; formulas follow SURVEY.md Appendix A7/A8 and are cross-checked in
tests/test_oracle_cpu.py (rotated IoU vs axis-aligned closed form at yaw in {0, pi/2} and Monte-Carlo areas).

Reference call sites: models/dense_heads/uni3detr_head.py:671 (nearest-BEV aligned), :695 (rotated 3-D, diag),
core/bbox/assigners/hungarian_assigner_3d.py:112, core/bbox/match_costs/match_cost.py:94, models/losses/rdiouloss.py:99.
"""
import math

import numpy as np
import torch


def limit_period(val, offset=0.5, period=math.pi):
    return val - torch.floor(val / period + offset) * period


def nearest_bev(boxes):
    """(x,y,z,dx,dy,dz,yaw) -> axis-aligned (x1,y1,x2,y2) of the yaw-snapped BEV box (mmdet3d BaseInstance3DBoxes.nearest_bev)."""
    bev = boxes[..., [0, 1, 3, 4, 6]]
    rot = torch.abs(limit_period(bev[..., -1], 0.5, math.pi))
    cond = (rot > math.pi / 4)[..., None]
    xywh = torch.where(cond, bev[..., [0, 1, 3, 2]], bev[..., :4])
    centers, dims = xywh[..., :2], xywh[..., 2:]
    return torch.cat([centers - dims / 2, centers + dims / 2], -1)


def bbox_overlaps_2d(b1, b2, is_aligned=False, eps=1e-6):
    """mmdet bbox_overlaps, mode='iou' on (x1,y1,x2,y2)."""
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if is_aligned:
        lt = torch.max(b1[..., :2], b2[..., :2])
        rb = torch.min(b1[..., 2:], b2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        ov = wh[..., 0] * wh[..., 1]
        union = a1 + a2 - ov
    else:
        lt = torch.max(b1[..., :, None, :2], b2[..., None, :, :2])
        rb = torch.min(b1[..., :, None, 2:], b2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        ov = wh[..., 0] * wh[..., 1]
        union = a1[..., None] + a2[..., None, :] - ov
    union = torch.max(union, union.new_tensor([eps]))
    return ov / union


def bbox_overlaps_nearest_3d(b1, b2, mode="iou", is_aligned=False, coordinate="lidar"):
    assert b1.size(-1) == b2.size(-1) >= 7
    return bbox_overlaps_2d(nearest_bev(b1[..., :7]), nearest_bev(b2[..., :7]), is_aligned)


# --------------------------------------------------------------------------------------------------
# rotated rectangle intersection (Sutherland-Hodgman, float64)
# --------------------------------------------------------------------------------------------------
def _corners(cx, cy, w, h, a):
    c, s = math.cos(a), math.sin(a)
    pts = []
    for sx, sy in ((-0.5, -0.5), (0.5, -0.5), (0.5, 0.5), (-0.5, 0.5)):
        x, y = sx * w, sy * h
        pts.append((cx + x * c - y * s, cy + x * s + y * c))
    return pts


def _clip(poly, a, b):
    out = []
    n = len(poly)
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def rotated_intersection_area(r1, r2):
    poly = _corners(*r1)
    clipper = _corners(*r2)
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, clipper[i], clipper[(i + 1) % 4])
    if len(poly) < 3:
        return 0.0
    area = 0.0
    for i in range(len(poly)):
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % len(poly)]
        area += x1 * y2 - x2 * y1
    return abs(area) * 0.5


def box_iou_rotated_aligned(bev1, bev2):
    """bev: [N,5] (cx,cy,w,h,angle) -> IoU [N] (mmcv.ops.box_iou_rotated, aligned)."""
    b1 = bev1.detach().double().cpu().numpy()
    b2 = bev2.detach().double().cpu().numpy()
    out = np.zeros((b1.shape[0],), np.float64)
    for i in range(b1.shape[0]):
        a1, a2 = b1[i, 2] * b1[i, 3], b2[i, 2] * b2[i, 3]
        if a1 < 1e-14 or a2 < 1e-14:
            continue
        inter = rotated_intersection_area(b1[i], b2[i])
        out[i] = inter / (a1 + a2 - inter)
    return torch.from_numpy(out).to(bev1.dtype)


def bbox_overlaps_3d_aligned(b1, b2):
    """diag(bbox_overlaps_3d(b1, b2, 'iou', coordinate='lidar')) — the only values the reference consumes
    (uni3detr_head.py:695).  Boxes are interpreted bottom-centre as upstream does."""
    bev1 = b1[:, [0, 1, 3, 4, 6]].clone()
    bev2 = b2[:, [0, 1, 3, 4, 6]].clone()
    bev1[:, 2:4] = bev1[:, 2:4].clamp(min=1e-4)
    bev2[:, 2:4] = bev2[:, 2:4].clamp(min=1e-4)
    iou2d = box_iou_rotated_aligned(bev1, bev2)
    areas1 = bev1[:, 2] * bev1[:, 3]
    areas2 = bev2[:, 2] * bev2[:, 3]
    overlaps_bev = iou2d * (areas1 + areas2) / (1 + iou2d)
    top = torch.min(b1[:, 2] + b1[:, 5], b2[:, 2] + b2[:, 5])
    bot = torch.max(b1[:, 2], b2[:, 2])
    overlaps_h = (top - bot).clamp(min=0)
    ov3d = overlaps_bev * overlaps_h
    v1 = b1[:, 3] * b1[:, 4] * b1[:, 5]
    v2 = b2[:, 3] * b2[:, 4] * b2[:, 5]
    return ov3d / torch.clamp(v1 + v2 - ov3d, min=1e-8)


def bbox_overlaps_3d(b1, b2, mode="iou", coordinate="lidar"):
    """Full [N,M] matrix (slow; only used by the reference shim on small inputs)."""
    n, m = b1.shape[0], b2.shape[0]
    i = torch.arange(n).repeat_interleave(m)
    j = torch.arange(m).repeat(n)
    return bbox_overlaps_3d_aligned(b1[i], b2[j]).view(n, m)
