"""Model sections of the other shipped Uni3DETR configurations, written as the handful of keys in which each differs from the
SUN RGB-D model (uni3detr_amd/configs/sunrgbd.py).  Used where the reference tree is absent (GPU box); tests/test_plugin_cpu.py
asserts they equal the `model` dicts of projects/configs/uni3detr/uni3detr_{scannet_large,kitti_3classes,nuscenes}.py when
those files are available."""
import copy

from .sunrgbd import model as _base


def _set(d, path, value):
    keys = path.split(".")
    for k in keys[:-1]:
        d = d[int(k)] if isinstance(d, (list, tuple)) else d[k]
    if value is _DELETE:
        d.pop(keys[-1], None)
    else:
        d[keys[-1]] = value


_DELETE = object()


def _variant(pc_range, overrides):
    m = copy.deepcopy(_base)
    for p in ("pts_voxel_layer.point_cloud_range", "pts_bbox_head.bbox_coder.pc_range", "pts_bbox_head.bbox_coder.post_center_range",
              "train_cfg.pts.assigner.pc_range", "train_cfg.pts.point_cloud_range"):
        _set(m, p, list(pc_range))
    for k, v in overrides.items():
        _set(m, k, v)
    return m


_SCANNET_RANGE = [-6.4, -6.4, -0.1, 6.4, 6.4, 2.46]
scannet_large = _variant(_SCANNET_RANGE, {
    "dynamic_voxelization": True,
    "pts_voxel_layer.max_num_points": -1, "pts_voxel_layer.max_voxels": (-1, -1),
    "pts_voxel_encoder": dict(type="DynamicSimpleVFE", voxel_size=[0.02, 0.02, 0.02], point_cloud_range=_SCANNET_RANGE),
    "pts_middle_encoder.sparse_shape": [128, 640, 640], "pts_middle_encoder.base_channels": 32, "pts_middle_encoder.output_channels": 512,
    "pts_middle_encoder.encoder_channels": ((32, 32, 64), (64, 64, 128), (128, 128, 256), (256, 256)),
    "pts_backbone.in_channels": [512, 512, 512],
    "pts_bbox_head.num_classes": 18, "pts_bbox_head.bbox_coder.num_classes": 18, "pts_bbox_head.bbox_coder.max_num": 5000,
    "train_cfg.pts.grid_size": [128, 640, 640],
})

_KITTI_RANGE = [0, -40, -3, 70.4, 40, 1]
_KITTI_VOXEL = [0.05, 0.05, 0.1]
kitti_3classes = _variant(_KITTI_RANGE, {
    "pts_voxel_layer.voxel_size": _KITTI_VOXEL, "pts_middle_encoder.sparse_shape": [41, 1600, 1408],
    "pts_bbox_head.num_classes": 3, "pts_bbox_head.gt_repeattimes": 5, "pts_bbox_head.transformer.decoder.num_layers": 9,
    "pts_bbox_head.bbox_coder.num_classes": 3, "pts_bbox_head.bbox_coder.max_num": 150, "pts_bbox_head.bbox_coder.alpha": 0.2,
    "pts_bbox_head.bbox_coder.voxel_size": _KITTI_VOXEL,
    "pts_bbox_head.post_processing": dict(type="box_merging", score_thr=[0.0, 0.3, 0.65]),
    "train_cfg.pts.grid_size": [1408, 1600, 40], "train_cfg.pts.voxel_size": _KITTI_VOXEL,
})

_NUS_RANGE = [-54, -54, -5.0, 54, 54, 3.0]
nuscenes = _variant(_NUS_RANGE, {
    "pts_voxel_layer.voxel_size": [0.075, 0.075, 0.2], "pts_voxel_layer.max_num_points": 10, "pts_voxel_layer.max_voxels": (90000, 120000),
    "pts_voxel_layer.deterministic": False,
    "pts_voxel_encoder.num_features": 5, "pts_middle_encoder.in_channels": 5, "pts_middle_encoder.sparse_shape": [41, 1440, 1440],
    "pts_bbox_head.num_query": 900, "pts_bbox_head.code_size": _DELETE, "pts_bbox_head.code_weights": [1.0] * 10,
    "pts_bbox_head.transformer.fp16_enabled": False,
    "pts_bbox_head.transformer.decoder.transformerlayers.attn_cfgs.1.fp16_enabled": False,
    "pts_bbox_head.bbox_coder.max_num": 900, "pts_bbox_head.bbox_coder.voxel_size": [0.15, 0.15, 8],
    "pts_bbox_head.bbox_coder.post_center_range": [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
    "pts_bbox_head.post_processing": dict(type="nms", nms_thr=0.2, num_thr=500),
    "train_cfg.pts.grid_size": [720, 720, 1], "train_cfg.pts.voxel_size": [0.15, 0.15, 8],
})

sunrgbd = copy.deepcopy(_base)
