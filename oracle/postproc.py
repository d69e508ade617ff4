"""ORACLE (test infrastructure): CPU restatement of the reference's test-time post-processing.

soft_nms      follows models/dense_heads/uni3detr_head.py:796-823 line by line (torch CPU; the rotated 3-D IoU is oracle/boxes.py's
              restatement of upstream bbox_overlaps_3d).
merge_boxes   follows core/bbox/bbox_merging.py:11-29 (boxes_3d_to_corners), :67-92 (overlapped_boxes_3d_fast_poly), :95-112
              (bboxes_sort), :115-145 (bboxes_nms_merge_only) as called from uni3detr_head.py:881-891.  That file imports cv2, shapely
              and numba at module level (none installed here: the file cannot be imported) - **parity unpinned** for shapely's
              polygon intersection, restated as a float64 Sutherland-Hodgman clip of two convex quadrilaterals (exact for convex
              polygons) and pinned by closed forms in tests/test_oracle_cpu.py.
"""
import numpy as np
import torch

from . import boxes as ob


def soft_nms(boxes, scores, gaussian_sigma=0.3, prune_threshold=1e-3):
    boxes, scores = boxes.clone(), scores.clone()
    idxs = torch.arange(scores.numel())
    out_i, out_s = [], []
    while scores.numel() > 0:
        top = int(torch.argmax(scores))
        out_i.append(int(idxs[top]))
        out_s.append(float(scores[top]))
        ious = ob.bbox_overlaps_3d(boxes[top:top + 1], boxes)[0]
        scores = scores * torch.exp(-ious.pow(2) / gaussian_sigma)
        keep = scores > prune_threshold
        keep[top] = False
        boxes, scores, idxs = boxes[keep], scores[keep], idxs[keep]
    return torch.tensor(out_i, dtype=torch.long), torch.tensor(out_s)


def soft_nms_classwise(boxes, scores, labels, num_classes, gaussian_sigma, prune_threshold):
    """the per-class loop of get_bboxes (:849-880): -> (indices into the input, decayed scores, labels), class-major."""
    oi, os_, ol = [], [], []
    for j in range(num_classes):
        ind = (labels == j).nonzero().reshape(-1)
        if ind.numel() == 0:
            continue
        ki, sc = soft_nms(boxes[ind][:, :7], scores[ind], gaussian_sigma, prune_threshold)
        oi.append(ind[ki]); os_.append(sc); ol.append(torch.full_like(ki, j))
    if not oi:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0), torch.zeros(0, dtype=torch.long)
    return torch.cat(oi), torch.cat(os_), torch.cat(ol)


def _corners(boxes):
    out = []
    for x3d, y3d, z3d, l, h, w, yaw in boxes.astype(np.float64):
        R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        c = np.array([[l / 2, 0.0, w / 2], [l / 2, 0.0, -w / 2], [-l / 2, 0.0, -w / 2], [-l / 2, 0.0, w / 2],
                      [l / 2, -h, w / 2], [l / 2, -h, -w / 2], [-l / 2, -h, -w / 2], [-l / 2, -h, w / 2]])
        out.append(c.dot(R.T) + np.array([x3d, y3d, z3d]))
    return np.array(out)


def _poly_area(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def _convex_intersection_area(p, q):
    """area of the intersection of two convex polygons given as [n,2] vertex loops (any orientation)."""
    def ccw(poly):
        x, y = poly[:, 0], poly[:, 1]
        return poly if (np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) >= 0 else poly[::-1]
    poly = [tuple(v) for v in ccw(p)]
    clip = [tuple(v) for v in ccw(q)]
    for i in range(len(clip)):
        if not poly:
            return 0.0
        poly = ob._clip(poly, clip[i], clip[(i + 1) % len(clip)])
    if len(poly) < 3:
        return 0.0
    return _poly_area(np.array(poly))


def _overlap(single, others):
    mx0, mn0 = single.max(0), single.min(0)
    mx, mn = others.max(1), others.min(1)
    out = np.zeros(len(others))
    non = np.any(np.logical_or(mx0 < mn, mn0 > mx), axis=1)
    p1 = single[:4][:, [0, 2]]
    a1 = _poly_area(p1)
    for i in range(len(others)):
        if non[i]:
            continue
        p2 = others[i][:4][:, [0, 2]]
        shared = _convex_intersection_area(p1, p2)
        a2 = _poly_area(p2)
        shared_y = min(mx[i][1], mx0[1]) - max(mn[i][1], mn0[1])
        inter = shared_y * shared
        union = (mx[i][1] - mn[i][1]) * a2 + (mx0[1] - mn0[1]) * a1
        out[i] = np.float32(inter) / (union - inter)
    return out


def merge_boxes(labels, boxes, scores, thr=0.1):
    """nms_boxes_3d_merge_only(labels, boxes, scores, overlapped_thres=thr, top_k=-1) -> (labels, merged boxes, scores, kept indices
    into the score-sorted order, sort order)."""
    order = np.argsort(-scores, kind="stable")
    labels, scores, boxes = labels[order], scores[order], boxes[order].copy()
    corners = _corners(boxes)
    keep = np.ones(scores.shape, dtype=bool)
    for i in range(scores.size - 1):
        if keep[i]:
            valid = keep[(i + 1):]
            ov = _overlap(corners[i], corners[(i + 1):][valid])
            rem = np.logical_and(ov > thr, labels[(i + 1):][valid] == labels[i])
            grp = np.concatenate([boxes[(i + 1):][valid][rem], boxes[[i]]], axis=0)
            boxes[i][:] = np.median(grp, axis=0)
            keep[(i + 1):][valid] = np.logical_not(rem)
    idx = np.where(keep)[0]
    return labels[idx], boxes[idx], scores[idx], idx, order
