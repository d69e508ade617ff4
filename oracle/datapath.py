"""CPU restatement (numpy, float32) of the training data-path transforms of SURVEY.md 8f-4 - TEST INFRASTRUCTURE ONLY: imported by
tests/ (and nothing in the product path).

Follows the shipped train pipeline (ref: projects/configs/uni3detr/uni3detr_sunrgbd.py:150-174: RandomFlip3D ->
GlobalRotScaleTrans -> PointsRangeFilter -> PointSample) and the plugin's Unified* transforms
(ref: projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py:325-589).

Pinning: the reference delegates the arithmetic to mmdet3d v1.0.0rc5's box / points classes, which are NOT under /root/reference and
cannot be imported here -> for `rotate` / `flip` / `scale` / `in_range_3d` / PointSample this restatement is written from the
upstream behaviour (recalled) = **parity unpinned** for those.  What the reference file itself states is pinned by
tests/test_oracle_cpu.py: the row-vector rotation matrix `rot_mat_T = [[c, s, 0], [-s, c, 0], [0, 0, 1]]` (:380-383, the
mmdet3d >= 1.0 branch), `uni_scale_mat = s * I` (:429-431), `flip_mat` with [1,1] negated for a horizontal and [0,0] for a vertical
flip (:571-580), and the composition `uni_rot_aug = flip_mat @ rot_mat_T @ uni_scale_mat` (:564-567 then :461-464): the augmented
xyz must equal `xyz @ uni_rot_aug` (Depth coordinates: a horizontal flip mirrors x there, which is flip_mat's convention only for
LiDAR - see uni_rot_aug()).
"""
import numpy as np

DEPTH, LIDAR = 0, 1


def rot_mat_T(angle):
    """transform_3d.py:380-383 (mmdet3d >= 1.0)."""
    s, c = np.float32(np.sin(np.float32(angle))), np.float32(np.cos(np.float32(angle)))
    return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], np.float32)


def flip_mat(flip_h, flip_v):
    """transform_3d.py:571-580: horizontal negates [1,1] (the y axis), vertical negates [0,0] (the x axis) - the LiDAR convention."""
    m = np.eye(3, dtype=np.float32)
    if flip_h:
        m[1, 1] *= -1
    if flip_v:
        m[0, 0] *= -1
    return m


def uni_rot_aug(flip_h, flip_v, angle, scale):
    """The matrix the Unified* transforms publish (:564-567, :461-464)."""
    return flip_mat(flip_h, flip_v) @ (rot_mat_T(angle) @ (np.eye(3, dtype=np.float32) * np.float32(scale)))


def _flip_xy(coord, flip_h, flip_v, x, y):
    """mmdet3d (recalled): Depth: horizontal mirrors x, vertical mirrors y; LiDAR: horizontal mirrors y, vertical mirrors x."""
    if coord == DEPTH:
        if flip_h:
            x = -x
        if flip_v:
            y = -y
    else:
        if flip_h:
            y = -y
        if flip_v:
            x = -x
    return x, y


def augment_points(points, flip_h, flip_v, angle, scale, coord=DEPTH, height_dim=-1, trans=(0.0, 0.0, 0.0)):
    """One scene: flip -> rotate (row vector @ rot_mat_T) -> scale -> translate (mmdet3d GlobalRotScaleTrans order, recalled); the height
    attribute scales when shift_height=True (:411-415)."""
    p = np.array(points, np.float32, copy=True)
    x, y = _flip_xy(coord, flip_h, flip_v, p[:, 0].copy(), p[:, 1].copy())
    s, c = np.float32(np.sin(np.float32(angle))), np.float32(np.cos(np.float32(angle)))
    sc = np.float32(scale)
    t = np.asarray(trans, np.float32)
    p[:, 0] = (x * c - y * s) * sc + t[0]
    p[:, 1] = (x * s + y * c) * sc + t[1]
    p[:, 2] = p[:, 2] * sc + t[2]
    if height_dim >= 3:
        p[:, height_dim] = p[:, height_dim] * sc
    return p


def augment_boxes(boxes, flip_h, flip_v, angle, scale, coord=DEPTH, trans=(0.0, 0.0, 0.0)):
    """Boxes (x, y, z, dx, dy, dz, yaw [, vx, vy]): centres like points, sizes scale, yaw: Depth h-flip pi - yaw, v-flip -yaw;
    LiDAR h-flip -yaw, v-flip pi - yaw; then + angle (mmdet3d v1.0 `rotate`, recalled); velocities flip / rotate and SCALE with the
    frame (mmdet3d BaseInstance3DBoxes.scale: `tensor[:, :6] *= s; tensor[:, 7:] *= s`, recalled)."""
    b = np.array(boxes, np.float32, copy=True)
    if b.shape[0] == 0:
        return b
    x, y = _flip_xy(coord, flip_h, flip_v, b[:, 0].copy(), b[:, 1].copy())
    yaw = b[:, 6].copy()
    pi = np.float32(np.pi)
    if coord == DEPTH:
        if flip_h:
            yaw = -yaw + pi
        if flip_v:
            yaw = -yaw
    else:
        if flip_h:
            yaw = -yaw
        if flip_v:
            yaw = -yaw + pi
    s, c = np.float32(np.sin(np.float32(angle))), np.float32(np.cos(np.float32(angle)))
    sc = np.float32(scale)
    t = np.asarray(trans, np.float32)
    b[:, 0] = (x * c - y * s) * sc + t[0]
    b[:, 1] = (x * s + y * c) * sc + t[1]
    b[:, 2] = b[:, 2] * sc + t[2]
    b[:, 3:6] = b[:, 3:6] * sc
    b[:, 6] = yaw + np.float32(angle)
    if b.shape[1] >= 9:
        vx, vy = _flip_xy(coord, flip_h, flip_v, b[:, 7].copy(), b[:, 8].copy())
        b[:, 7] = (vx * c - vy * s) * sc
        b[:, 8] = (vx * s + vy * c) * sc
    return b


def object_range_filter(boxes, labels, pc_range):
    """ObjectRangeFilter (mmdet3d, recalled): in_range_bev on the box centres (strict), labels follow, limit_yaw(0.5, 2 pi)."""
    b = np.array(boxes, np.float32, copy=True)
    x0, y0, x1, y1 = (np.float32(pc_range[i]) for i in (0, 1, 3, 4))
    keep = (b[:, 0] > x0) & (b[:, 1] > y0) & (b[:, 0] < x1) & (b[:, 1] < y1)
    b = b[keep]
    two_pi = np.float32(6.283185307179586)
    b[:, 6] = b[:, 6] - np.floor(b[:, 6] / two_pi + np.float32(0.5)) * two_pi
    return b, np.asarray(labels)[keep]


def range_filter(points, pc_range):
    """PointsRangeFilter (mmdet3d `in_range_3d`, recalled): strict inequalities, order kept."""
    p = np.asarray(points, np.float32)
    lo, hi = np.asarray(pc_range[:3], np.float32), np.asarray(pc_range[3:], np.float32)
    keep = (p[:, 0] > lo[0]) & (p[:, 1] > lo[1]) & (p[:, 2] > lo[2]) & (p[:, 0] < hi[0]) & (p[:, 1] < hi[1]) & (p[:, 2] < hi[2])
    return p[keep]


def point_sample(points, num_points, rng):
    """PointSample (mmdet3d `_points_random_sampling` without sample_range, recalled):
    choices = rng.choice(n, num_points, replace = n < num_points)."""
    n = len(points)
    if n == 0:
        return np.zeros((num_points, points.shape[1]), np.float32), np.full(num_points, -1, np.int64)
    idx = rng.choice(n, num_points, replace=n < num_points)
    return np.asarray(points, np.float32)[idx], idx
