"""On-device training data path: the transforms of the shipped train pipelines applied to a packed batch that is
already resident in HBM, so that real-data epochs are not bound by DataLoader workers.

Mirrors the pipeline entries the configs name (ref: projects/configs/uni3detr/uni3detr_sunrgbd.py:150-174 and the plugin's
Unified* variants, projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py:325-589): same `type` names, same constructor
arguments, same result-dict keys (`pcd_horizontal_flip`, `pcd_vertical_flip`, `pcd_rotation_angle`, `pcd_scale_factor`,
`uni_rot_aug`), one call for the whole BATCH instead of one per sample.  The random draws are host-side numpy draws exactly as in
the reference (`np.random.rand() < ratio`, `np.random.uniform(lo, hi)`: one per scene, in scene order, flip transforms first), the
arithmetic runs in libu3d_hip.so (uni3detr_amd/csrc/datapath.hip) - there is no host/torch fallback.

A batch is a dict: points [N,F] f32 (all scenes packed), scene_off int32 [B+1] (device), optional count int32 [B] (live rows at
the front of every scene's segment, set by PointsRangeFilter), gt_bboxes_3d [G,7|9] f32 packed + gt_off int32 [B+1] (device).
"""
import numpy as np
import torch

from . import native as nv
from .registry import Registry

PIPELINES = Registry("pipeline")
DEPTH, LIDAR = 0, 1


def _coord(batch):
    return LIDAR if str(batch.get("box_type_3d", "Depth")).lower().startswith("lidar") else DEPTH


def _params(batch):
    """Device parameter table [B,9] = (flip_h, flip_v, sin, cos, angle, scale, tx, ty, tz) from the draws recorded in the batch dict."""
    B = batch["scene_off"].numel() - 1
    fh = np.asarray(batch.get("pcd_horizontal_flip", np.zeros(B, bool)), np.float32)
    fv = np.asarray(batch.get("pcd_vertical_flip", np.zeros(B, bool)), np.float32)
    ang = np.asarray(batch.get("pcd_rotation_angle", np.zeros(B)), np.float32)
    sc = np.asarray(batch.get("pcd_scale_factor", np.ones(B)), np.float32)
    tr = np.asarray(batch.get("pcd_trans", np.zeros((B, 3))), np.float32).reshape(B, 3)
    tab = np.concatenate([np.stack([fh, fv, np.sin(ang), np.cos(ang), ang, sc], 1), tr], 1).astype(np.float32)
    return torch.from_numpy(tab).to(batch["points"].device)


def _apply(batch, fh, fv, ang, sc, height_dim, trans=None):
    """One launch over the points (+ one over the boxes) for the given per-scene draws; identity entries cost nothing extra."""
    B = batch["scene_off"].numel() - 1
    tmp = dict(scene_off=batch["scene_off"], points=batch["points"], pcd_horizontal_flip=fh, pcd_vertical_flip=fv,
               pcd_rotation_angle=ang, pcd_scale_factor=sc, pcd_trans=np.zeros((B, 3), np.float32) if trans is None else trans)
    tab = _params(tmp)
    coord = _coord(batch)
    nv.points_augment(batch["points"], batch["scene_off"], tab, coord, height_dim)
    g = batch.get("gt_bboxes_3d")
    if g is not None and g.shape[0] > 0:
        nv.boxes_augment(g, batch["gt_off"], tab, coord)
    # the matrix the Unified* transforms publish (transform_3d.py:461-464, :564-567), per scene, composed with earlier transforms
    mats = []
    for b in range(B):
        s, c = np.float32(np.sin(np.float32(ang[b]))), np.float32(np.cos(np.float32(ang[b])))
        flip = np.eye(3, dtype=np.float32)
        if fh[b]:
            flip[1, 1] *= -1
        if fv[b]:
            flip[0, 0] *= -1
        m = flip @ (np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], np.float32) @ (np.eye(3, dtype=np.float32) * np.float32(sc[b])))
        prev = batch.get("uni_rot_aug")
        mats.append(m if prev is None else prev[b] @ m)
    batch["uni_rot_aug"] = mats
    return batch


@PIPELINES.register_module()
class RandomFlip3D:
    """ref: mmdet3d RandomFlip3D as configured at uni3detr_sunrgbd.py:159-163 (sync_2d has no effect without images)."""

    def __init__(self, sync_2d=True, flip_ratio_bev_horizontal=0.0, flip_ratio_bev_vertical=0.0, **kwargs):
        assert 0 <= flip_ratio_bev_horizontal <= 1 and 0 <= flip_ratio_bev_vertical <= 1
        self.flip_ratio_bev_horizontal, self.flip_ratio_bev_vertical = flip_ratio_bev_horizontal, flip_ratio_bev_vertical

    def __call__(self, batch):
        B = batch["scene_off"].numel() - 1
        # the reference draws per sample: horizontal first, then vertical (transform_3d.py:552-559)
        if "pcd_horizontal_flip" not in batch or "pcd_vertical_flip" not in batch:
            fh, fv = np.zeros(B, bool), np.zeros(B, bool)
            for b in range(B):
                fh[b] = np.random.rand() < self.flip_ratio_bev_horizontal
                fv[b] = np.random.rand() < self.flip_ratio_bev_vertical
            batch.setdefault("pcd_horizontal_flip", fh)
            batch.setdefault("pcd_vertical_flip", fv)
        fh, fv = np.asarray(batch["pcd_horizontal_flip"], bool), np.asarray(batch["pcd_vertical_flip"], bool)
        batch.setdefault("transformation_3d_flow", []).extend(["HF"] * int(fh.any()) + ["VF"] * int(fv.any()))
        return _apply(batch, fh, fv, np.zeros(B, np.float32), np.ones(B, np.float32), -1)


@PIPELINES.register_module()
class UnifiedRandomFlip3D(RandomFlip3D):
    """ref: transform_3d.py:486-589 (same geometry; `uni_rot_aug` is published by both classes here)."""


@PIPELINES.register_module()
class GlobalRotScaleTrans:
    """ref: mmdet3d GlobalRotScaleTrans as configured at uni3detr_sunrgbd.py:164-168 and, with translation_std = [.1, .1, .1], in the
    two ScanNet configs (uni3detr_scannet.py / uni3detr_scannet_large.py train_pipeline): rotate -> scale -> translate, the translation
    one normal draw per axis and scene added to the points and to the box centres (upstream `_trans_bbox_points`, recalled)."""

    def __init__(self, rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05), translation_std=(0, 0, 0), shift_height=False):
        if not isinstance(rot_range, (list, tuple, np.ndarray)):
            rot_range = [-rot_range, rot_range]
        if not isinstance(translation_std, (list, tuple, np.ndarray)):
            translation_std = [translation_std] * 3
        assert len(translation_std) == 3 and all(t >= 0 for t in translation_std), "invalid translation_std"
        self.translation_std = np.asarray(translation_std, np.float32)
        self.rot_range, self.scale_ratio_range, self.shift_height = list(rot_range), list(scale_ratio_range), shift_height

    def __call__(self, batch):
        B = batch["scene_off"].numel() - 1
        if "pcd_rotation_angle" not in batch:        # per sample: rotation first, then scale, then translation (transform_3d.py:456-460)
            ang, sc, tr = np.zeros(B, np.float32), np.ones(B, np.float32), np.zeros((B, 3), np.float32)
            for b in range(B):
                ang[b] = np.random.uniform(self.rot_range[0], self.rot_range[1])
                sc[b] = np.random.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1])
                if np.any(self.translation_std != 0):
                    tr[b] = np.random.normal(scale=self.translation_std, size=3)
            batch["pcd_rotation_angle"] = ang
            batch.setdefault("pcd_scale_factor", sc)
            batch.setdefault("pcd_trans", tr)
        ang = np.asarray(batch["pcd_rotation_angle"], np.float32)
        sc = np.asarray(batch.get("pcd_scale_factor", np.ones(B)), np.float32)
        tr = np.asarray(batch.get("pcd_trans", np.zeros((B, 3))), np.float32).reshape(B, 3)
        hd = int(batch.get("height_dim", 3)) if self.shift_height else -1
        batch.setdefault("transformation_3d_flow", []).extend(["R", "S", "T"])
        return _apply(batch, np.zeros(B, bool), np.zeros(B, bool), ang, sc, hd, tr)


@PIPELINES.register_module()
class UnifiedRotScaleTrans(GlobalRotScaleTrans):
    """ref: transform_3d.py:326-483."""

    def __init__(self, rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05), shift_height=False):
        super().__init__(rot_range, scale_ratio_range, (0, 0, 0), shift_height)


@PIPELINES.register_module()
class PointsRangeFilter:
    """ref: uni3detr_sunrgbd.py:169 (mmdet3d PointsRangeFilter): survivors keep their order; every scene's segment keeps its offset,
    `count` says how many rows at its front are live."""

    def __init__(self, point_cloud_range):
        self.pcd_range = [float(v) for v in point_cloud_range]

    def __call__(self, batch):
        out, count = nv.points_range_filter(batch["points"], batch["scene_off"], self.pcd_range, out=batch["points"])
        batch["points"], batch["count"] = out, count
        return batch


@PIPELINES.register_module()
class PointSample:
    """ref: uni3detr_sunrgbd.py:171 (mmdet3d PointSample): every scene becomes exactly num_points rows."""

    def __init__(self, num_points, sample_range=None, replace=False):
        if sample_range is not None:
            raise NotImplementedError("sample_range is not used by any shipped Uni3DETR config")
        self.num_points = int(num_points)
        self._seed = None

    def __call__(self, batch):
        dev = batch["points"].device
        if self._seed is None or self._seed.device != dev:
            self._seed = torch.tensor([int(np.random.randint(0, 2 ** 62))], dtype=torch.int64, device=dev)
        else:
            self._seed += 0x9E3779B97F4A7C15 - (1 << 64)       # a new stream every call, device-side (capturable)
        B = batch["scene_off"].numel() - 1
        batch["points"] = nv.point_sample(batch["points"], batch["scene_off"], batch.get("count"), self.num_points, self._seed)
        batch["scene_off"] = torch.arange(0, (B + 1) * self.num_points, self.num_points, dtype=torch.int32, device=dev)
        batch.pop("count", None)
        return batch


@PIPELINES.register_module()
class ObjectRangeFilter:
    """ref: uni3detr_kitti_3classes.py / uni3detr_nuscenes.py train_pipeline (mmdet3d ObjectRangeFilter, recalled): ground-truth boxes
    whose BEV centre left (x0, y0, x1, y1) after flip / rotation / scale are dropped together with their labels, yaw is wrapped into
    [-pi, pi).  It runs AFTER the geometric augmentation, so with the augmentation on the device it has to be a device transform too:
    per scene, in place, order kept; `gt_count` says how many rows at the front of every scene's segment are live."""

    def __init__(self, point_cloud_range):
        r = [float(v) for v in point_cloud_range]
        self.bev_range = [r[0], r[1], r[3], r[4]]

    def __call__(self, batch):
        g = batch.get("gt_bboxes_3d")
        if g is None:
            return batch
        lab = batch.get("gt_labels_3d")
        if lab is not None and lab.dtype != torch.int32:
            lab = batch["gt_labels_3d"] = lab.to(torch.int32)
        if g.shape[0] == 0:
            # a batch without a single GT box (plausible for KITTI / nuScenes at 2-4 scenes per GPU): nothing to filter, and an empty
            # tensor has no device pointer to hand to the kernel
            batch["gt_count"] = torch.zeros(batch["gt_off"].numel() - 1, dtype=torch.int32, device=batch["gt_off"].device)
            return batch
        batch["gt_count"] = nv.boxes_range_filter(g, lab, batch["gt_off"], self.bev_range)
        return batch


@PIPELINES.register_module()
class MultiScaleFlipAug3D:
    """Test-time wrapper (mmdet3d): one scale, no flip in the plugin's use - the inner transforms run as they are."""

    def __init__(self, transforms, img_scale=None, pts_scale_ratio=1, flip=False, **kwargs):
        if flip or (pts_scale_ratio not in (1, 1.0, [1], [1.0])):
            raise NotImplementedError("test-time flips / point scaling are not used by any shipped Uni3DETR config")
        self.inner = DevicePipeline(transforms)

    def __call__(self, batch):
        return self.inner(batch)


_PASSTHROUGH = {"LoadPointsFromFile", "LoadAnnotations3D", "DefaultFormatBundle3D", "Collect3D", "CollectUnified3D", "LoadPointsFromMultiSweeps",
                "ObjectNameFilter", "PointShuffle", "ObjectSample", "UnifiedObjectSample", "ObjectNoise", "NormalizePointsColor",
                "LoadImageFromFile", "LoadMultiViewImageFromFiles"}


class DevicePipeline:
    """The device-side part of a config's `train_pipeline` / `test_pipeline` list: loading / formatting entries (and the
    ground-truth database sampler, which needs the dataset's files) stay with the host loader and are skipped here."""

    def __init__(self, pipeline_cfg):
        self.transforms, self.skipped = [], []
        for c in pipeline_cfg:
            if c["type"] in PIPELINES:
                self.transforms.append(PIPELINES.build(c))
            elif c["type"] in _PASSTHROUGH:
                self.skipped.append(c["type"])
            else:
                raise KeyError(f"pipeline entry {c['type']!r}: neither a device transform nor a known host-side entry")

    def __call__(self, batch):
        for t in self.transforms:
            batch = t(batch)
        return batch


def pack_batch(points, gt_bboxes_3d=None, box_type_3d="Depth", height_dim=3, gt_labels_3d=None):
    """list of per-scene [n_i,F] tensors (+ list of [g_i,7|9] box tensors, + list of label tensors) on one device -> the batch dict
    the transforms take."""
    dev = points[0].device
    lens = [int(p.shape[0]) for p in points]
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=dev)
    batch = dict(points=torch.cat([p.float() for p in points]).contiguous(), scene_off=off, box_type_3d=box_type_3d, height_dim=height_dim)
    if gt_bboxes_3d is not None:
        gl = [int(g.shape[0]) for g in gt_bboxes_3d]
        batch["gt_off"] = torch.tensor(np.concatenate([[0], np.cumsum(gl)]).astype(np.int32), device=dev)
        dim = gt_bboxes_3d[0].shape[1] if len(gt_bboxes_3d) else 7
        batch["gt_bboxes_3d"] = (torch.cat([g.float() for g in gt_bboxes_3d]).contiguous() if sum(gl)
                                 else torch.zeros((0, dim), dtype=torch.float32, device=dev))
        if gt_labels_3d is not None:
            batch["gt_labels_3d"] = (torch.cat([l.to(torch.int32) for l in gt_labels_3d]).contiguous() if sum(gl)
                                     else torch.zeros((0,), dtype=torch.int32, device=dev))
    return batch


def unpack_batch(batch, labels=None):
    """batch dict -> (points list, Boxes3D list[, labels list]) as `Uni3DETR.forward_train` / `TrainStep.set_batch` take them
    (views of the packed tensors: no copies)."""
    from .plugin.structures import Boxes3D
    off = batch["scene_off"].tolist()
    pts = [batch["points"][off[b]:off[b + 1]] for b in range(len(off) - 1)]
    if "count" in batch:
        cnt = batch["count"].tolist()
        pts = [p[:c] for p, c in zip(pts, cnt)]
    out = [pts]
    if "gt_bboxes_3d" in batch:
        go = batch["gt_off"].tolist()
        gc = batch["gt_count"].tolist() if "gt_count" in batch else [go[b + 1] - go[b] for b in range(len(go) - 1)]     # ObjectRangeFilter survivors
        out.append([Boxes3D(batch["gt_bboxes_3d"][go[b]:go[b] + gc[b]]) for b in range(len(go) - 1)])
        if labels is None and "gt_labels_3d" in batch:
            labels = [batch["gt_labels_3d"][go[b]:go[b] + gc[b]].long() for b in range(len(go) - 1)]
    if labels is not None:
        out.append(labels)
    return tuple(out)
