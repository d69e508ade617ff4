"""GPU: the other shipped configurations (BASELINE.json configs[2..4]) build from the reference config files when present
(else from restated key numbers), run one training step on scaled synthetic clouds and produce finite losses and gradients.
nuScenes is shape-only for the forward (its loss path is inconsistent upstream: SURVEY.md App. D-15)."""
import os

import numpy as np
import pytest
import torch

import projects.mmdet3d_plugin  # noqa: F401
from uni3detr_amd.plugin.structures import Boxes3D
from uni3detr_amd.registry import Config, build_model
from uni3detr_amd.synth import room_scene

pytestmark = pytest.mark.gpu
REF_CFG = "/root/reference/projects/configs/uni3detr"


def _cfg(name):
    import copy
    from uni3detr_amd.configs import variants
    p = os.path.join(REF_CFG, f"uni3detr_{name}.py")
    if os.path.exists(p):
        return Config.fromfile(p).model
    return copy.deepcopy(getattr(variants, name))


def _scene(i, n, rng_range, nfeat):
    p, g, l = room_scene(i, n, pc_range=rng_range)
    if nfeat > 4:
        p = np.concatenate([p, np.zeros((p.shape[0], nfeat - 4), np.float32)], 1)
    gb = torch.from_numpy(g).clone()
    gb[:, 2] -= gb[:, 5] / 2
    return torch.from_numpy(p), gb, torch.from_numpy(l)


@pytest.mark.parametrize("name,npts,bf16", [("kitti_3classes", 20000, True), ("scannet_large", 60000, True), ("sunrgbd", 20000, True)])
def test_config_trains_one_step(cuda, name, npts, bf16):
    cfg = _cfg(name)
    model = build_model(cfg).to(cuda).train()
    if bf16:
        model.set_precision("bf16")
    rng_range = tuple(cfg["pts_voxel_layer"]["point_cloud_range"])
    nfeat = cfg["pts_middle_encoder"]["in_channels"]
    ncls = cfg["pts_bbox_head"]["num_classes"]
    B = 2
    sc = [_scene(i, npts - 1000 * i, rng_range, nfeat) for i in range(B)]
    losses = model(return_loss=True, points=[s[0].to(cuda) for s in sc], img_metas=None,
                   gt_bboxes_3d=[Boxes3D(s[1]).to(cuda) for s in sc], gt_labels_3d=[(s[2] % ncls).to(cuda) for s in sc])
    assert len(losses) == 4 * cfg["pts_bbox_head"]["transformer"]["decoder"]["num_layers"]
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    for n_, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n_


def test_nuscenes_forward_shapes(cuda):
    cfg = _cfg("nuscenes")
    model = build_model(cfg).to(cuda).train().set_precision("bf16")
    rng_range = tuple(cfg["pts_voxel_layer"]["point_cloud_range"])
    sc = [_scene(i, 60000, rng_range, 5) for i in range(2)]
    feat, fps = model.extract_pts_feat([s[0].to(cuda) for s in sc])
    assert tuple(feat.shape) == (2, 256, 5, 180, 180) and tuple(fps.shape) == (2, 1800, 3)
    outs = model.pts_bbox_head(feat.requires_grad_(True), None, fps)
    assert tuple(outs["all_cls_scores"].shape) == (3, 2, 2700, 10) and tuple(outs["all_bbox_preds"].shape) == (3, 2, 2700, 10)


def test_inference_path_runs(cuda):
    cfg = _cfg("sunrgbd")
    model = build_model(cfg).to(cuda).eval()
    sc = [_scene(i, 20000, tuple(cfg["pts_voxel_layer"]["point_cloud_range"]), 4) for i in range(2)]
    res = model(return_loss=False, img_metas=[[dict(), dict()]], points=[s[0].to(cuda) for s in sc])
    assert len(res) == 2
    for r in res:
        n = r["boxes_3d"].shape[0]
        assert r["boxes_3d"].shape[1] == 7 and r["scores_3d"].shape == (n,) and r["labels_3d"].shape == (n,)
        assert (r["scores_3d"] >= 0).all() and (r["scores_3d"] <= 1).all()
