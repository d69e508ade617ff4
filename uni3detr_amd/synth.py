"""Seeded synthetic SUN-RGB-D-shaped scenes: points on room surfaces + 8 boxes, GT boxes/labels.

Host-side numpy only; the bench uploads the result once and keeps it resident in HBM.
"""
import numpy as np

SUNRGBD_RANGE = (-3.2, -0.2, -2.0, 3.2, 6.2, 0.56)
SUNRGBD_VOXEL = (0.02, 0.02, 0.02)


def room_scene(scene_id, n_points=20000, n_boxes=8, num_classes=10, pc_range=SUNRGBD_RANGE, seed_base=1234):
    """Returns points f32 [n_points,4] (x,y,z,height), gt f32 [n_boxes,7] (cx,cy,cz_gravity,dx,dy,dz,yaw), labels i64."""
    rng = np.random.default_rng(seed_base + scene_id)
    x0, y0, z0, x1, y1, z1 = pc_range
    sx, sy = (x1 - x0), (y1 - y0)
    # scale the canonical 6x6 m room to the configured range (identity for SUN RGB-D)
    fx, fy = sx / 6.4, sy / 6.4
    floor_z = z0 + 0.1
    surfaces = []  # (origin, u, v, area)
    rx0, rx1, ry0, ry1 = x0 + 0.2 * fx, x0 + 6.2 * fx, y0 + 0.2 * fy, y0 + 6.2 * fy
    surfaces.append((np.array([rx0, ry0, floor_z]), np.array([rx1 - rx0, 0, 0]), np.array([0, ry1 - ry0, 0])))
    wall_h = (z1 - z0) * 0.9
    surfaces.append((np.array([rx0, ry1, floor_z]), np.array([rx1 - rx0, 0, 0]), np.array([0, 0, wall_h])))
    surfaces.append((np.array([rx0, ry0, floor_z]), np.array([0, ry1 - ry0, 0]), np.array([0, 0, wall_h])))
    gt = np.zeros((n_boxes, 7), np.float32)
    for b in range(n_boxes):
        cx = rng.uniform(rx0 + 0.7 * fx, rx1 - 0.7 * fx)
        cy = rng.uniform(ry0 + 0.7 * fy, ry1 - 0.7 * fy)
        dx = rng.uniform(0.4, 2.0) * min(fx, 1.0)
        dy = rng.uniform(0.4, 1.2) * min(fy, 1.0)
        dz = min(rng.uniform(0.4, 1.2), wall_h * 0.8)
        yaw = rng.uniform(-np.pi, np.pi)
        cz = floor_z + dz / 2
        gt[b] = (cx, cy, cz, dx, dy, dz, yaw)
        c, s = np.cos(yaw), np.sin(yaw)
        ux, uy = np.array([c, s, 0.0]) * dx, np.array([-s, c, 0.0]) * dy
        uz = np.array([0, 0, dz])
        o = np.array([cx, cy, cz]) - ux / 2 - uy / 2 - uz / 2
        surfaces += [(o + uz, ux, uy), (o, ux, uz), (o + uy, ux, uz), (o, uy, uz), (o + ux, uy, uz)]
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in surfaces])
    counts = rng.multinomial(n_points, areas / areas.sum())
    pts = []
    for (o, u, v), k in zip(surfaces, counts):
        a, b = rng.random(k), rng.random(k)
        pts.append(o[None] + a[:, None] * u[None] + b[:, None] * v[None])
    p = np.concatenate(pts) + rng.normal(0, 0.005, (n_points, 3))
    p = p[rng.permutation(n_points)]
    h = p[:, 2:3] - p[:, 2].min()
    points = np.concatenate([p, h], 1).astype(np.float32)
    labels = rng.integers(0, num_classes, n_boxes).astype(np.int64)
    return points, gt, labels


def uniform_scene(scene_id, n_points=20000, pc_range=SUNRGBD_RANGE, seed_base=4321):
    """Uniform-in-volume cloud: the hash / active-set worst case."""
    rng = np.random.default_rng(seed_base + scene_id)
    lo, hi = np.array(pc_range[:3]), np.array(pc_range[3:])
    p = lo + rng.random((n_points, 3)) * (hi - lo)
    h = p[:, 2:3] - p[:, 2].min()
    return np.concatenate([p, h], 1).astype(np.float32)
