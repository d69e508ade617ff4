"""CPU: host logic of the drop-in layer — config loading, registry surface, checkpoint key names, C-ABI exports."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import projects.mmdet3d_plugin  # noqa: F401  (registers everything, like plugin_dir= in the shipped configs)
from uni3detr_amd import native as nv
from uni3detr_amd.registry import (ATTENTION, BACKBONES, BBOX_ASSIGNERS, BBOX_CODERS, Config, DETECTORS, HEADS, LOSSES, MATCH_COST,
                                   MIDDLE_ENCODERS, NECKS, TRANSFORMER, TRANSFORMER_LAYER_SEQUENCE, build_model)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = "/root/reference/projects/configs/uni3detr"
G = os.path.join(os.path.dirname(__file__), "golden")


def test_registry_surface():
    for reg, names in [(DETECTORS, ["Uni3DETR"]), (MIDDLE_ENCODERS, ["SparseEncoderHD"]), (BACKBONES, ["SECOND3D"]),
                       (NECKS, ["SECOND3DFPN"]), (HEADS, ["Uni3DETRHead"]), (TRANSFORMER, ["Uni3DETRTransformer"]),
                       (TRANSFORMER_LAYER_SEQUENCE, ["Uni3DETRTransformerDecoder"]), (ATTENTION, ["UniCrossAtten", "MultiheadAttention"]),
                       (BBOX_ASSIGNERS, ["HungarianAssigner3D"]), (MATCH_COST, ["BBox3DL1Cost", "IoU3DCost", "FocalLossCost"]),
                       (BBOX_CODERS, ["NMSFreeCoder"]), (LOSSES, ["SoftFocalLoss", "IoU3DLoss", "L1Loss"])]:
        for n in names:
            assert n in reg, (reg.name, n)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "u3d_hip.h")).read()
    declared = set(re.findall(r"\b(u3d_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"u3d_stream"}
    lib = ctypes.CDLL(nv.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/u3d_hip.h but not exported"
    assert declared == set(nv.exported_symbols()), declared ^ set(nv.exported_symbols())
    assert lib.u3d_version() >= 1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(nv, "_lib", None)
    monkeypatch.setattr(nv, "LIB_PATH", "/nonexistent/libu3d_hip.so")
    with pytest.raises(nv.U3DError):
        nv.lib()


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference configs only exist in the build container")
@pytest.mark.parametrize("name", ["sunrgbd", "scannet", "scannet_large", "kitti_3classes", "kitti_car", "nuscenes"])
def test_shipped_configs_load_unchanged_and_build(name):
    cfg = Config.fromfile(os.path.join(REF_CFG, f"uni3detr_{name}.py"))
    assert cfg.model.type == "Uni3DETR" and cfg.plugin_dir == "projects/mmdet3d_plugin/"
    assert cfg.dist_params.backend == "nccl"            # came from the supplied _base_/default_runtime.py
    model = build_model(cfg.model)
    assert model.pts_bbox_head.num_query == cfg.model.pts_bbox_head.num_query
    cfg.merge_from_dict({"model.pts_bbox_head.num_query": 10})
    assert cfg.model.pts_bbox_head.num_query == 10


def test_builtin_sunrgbd_config_equals_shipped():
    from uni3detr_amd.configs.sunrgbd import model as mine
    if os.path.isdir(REF_CFG):
        ref = Config.fromfile(os.path.join(REF_CFG, "uni3detr_sunrgbd.py")).model
        import json
        assert json.dumps(mine, sort_keys=True, default=list) == json.dumps(ref, sort_keys=True, default=list)
    m = build_model(mine)
    assert sum(p.numel() for p in m.parameters()) == 31707880


def test_head_state_dict_keys_match_reference_checkpoint_names():
    from uni3detr_amd.configs.sunrgbd import model as mine
    m = build_model(mine)
    z = np.load(os.path.join(G, "head_train_b2.npz"))
    mine_keys = {k: tuple(v.shape) for k, v in m.pts_bbox_head.state_dict().items()}
    ref_keys = {str(k): tuple(int(s) for s in str(shp).split(",")) if str(shp) else () for k, shp in zip(z["sd_names"], z["sd_shapes"])}
    assert mine_keys == ref_keys
    sd = m.state_dict()
    for k in ["pts_middle_encoder.conv_input.0.weight", "pts_middle_encoder.encoder_layers.encoder_layer1.0.conv1.weight",
              "pts_middle_encoder.encoder_layers.encoder_layer1.2.0.weight", "pts_middle_encoder.encoder_layers.encoder_layer4.1.bn2.running_var",
              "pts_middle_encoder.conv_out.1.num_batches_tracked", "pts_backbone.blocks.2.15.weight", "pts_backbone.blocks.0.16.bias",
              "pts_neck.deblocks.1.0.weight", "pts_neck.extra_blocks.6.weight", "pts_neck.extra_blocks.7.running_mean",
              "pts_bbox_head.transformer.decoder.layers.0.attentions.1.output_proj.weight", "pts_bbox_head.code_weights"]:
        assert k in sd, k
    assert tuple(sd["pts_middle_encoder.encoder_layers.encoder_layer3.2.0.weight"].shape) == (3, 3, 3, 64, 128)
    assert tuple(sd["pts_neck.deblocks.2.0.weight"].shape) == (512, 256, 1, 4, 4)


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference configs only exist in the build container")
@pytest.mark.parametrize("name", ["scannet_large", "kitti_3classes", "nuscenes", "sunrgbd"])
def test_builtin_variants_equal_shipped_configs(name):
    import json
    from uni3detr_amd.configs import variants
    ref = Config.fromfile(os.path.join(REF_CFG, f"uni3detr_{name}.py")).model
    assert json.dumps(getattr(variants, name), sort_keys=True, default=list) == json.dumps(ref, sort_keys=True, default=list)
