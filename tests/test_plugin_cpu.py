"""CPU: host logic of the drop-in layer — config loading, registry surface, checkpoint key names, C-ABI exports."""
import copy
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


def test_hot_kernels_keep_their_register_budget(tmp_path):
    """Guard against silent register-allocation regressions in the built library (one refactor of the LDS-DMA kernels' epilogue made
    the 256x256 kernel spill and halved the 128x128 kernel's occupancy: 25.4 -> 27.5 ms per step with every test green).  Reads the
    gfx950 code objects' metadata: no scratch in the implicit-GEMM kernels; the 256x256 tiles fit two waves per SIMD (<= 256 VGPRs),
    the 128x128 forward tile two workgroups per CU (<= 256 VGPR + AGPR)."""
    import re
    import shutil
    import subprocess
    from uni3detr_amd import native as nv
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(nv.LIB_PATH)):
        pytest.skip("ROCm LLVM tools or the built library are not available")
    lib = shutil.copy(nv.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, check=True, capture_output=True)
    kernels = {}
    for co in sorted(tmp_path.glob("lib.so.*gfx950")):
        notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
            kernels[name.group(1)] = dict(agpr=int(blk.split()[0]), vgpr=get("vgpr_count"), scratch=get("private_segment_fixed_size"))
    find = lambda sub: [v for k, v in kernels.items() if sub in k]
    assert len(kernels) > 50
    for sub in ("k_igemm_glds_256x256", "k_igemm_glds_128x128", "k_igemm_glds_128x64", "k_igemm_wgrad_glds_256", "k_igemm_wgrad_glds_128",
                "k_igemm_glds8_256x256", "k_igemm_glds8_256x256_f32o", "k_igemm_glds8_256x128_f32o"):
        ks = find(sub)
        assert ks, sub
        assert all(k["scratch"] == 0 for k in ks), (sub, ks)
    assert all(k["vgpr"] <= 256 and k["agpr"] == 0 for k in find("k_igemm_glds_256x256"))
    assert all(k["vgpr"] + k["agpr"] <= 256 for k in find("k_igemm_glds_128x128"))


def test_nms_free_coder_decode_matches_reference_golden():
    """NMSFreeCoder.decode vs the reference file's own output (tests/golden/coder_decode.npz, generated by oracle/make_golden.py from
    core/bbox/coders/nms_free_coder.py:42-136): top-k over query x class, score^alpha * iou^(1-alpha), score and centre-range masks."""
    import numpy as np
    from uni3detr_amd.plugin.bbox import NMSFreeCoder
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "coder_decode.npz"))
    preds = dict(all_cls_scores=torch.from_numpy(z["cls"]), all_bbox_preds=torch.from_numpy(z["box"]), all_iou_preds=torch.from_numpy(z["iou"]))
    si = 0
    while f"s{si}_cfg" in z:
        c = z[f"s{si}_cfg"]
        coder = NMSFreeCoder(pc_range=list(z["pc_range"]), voxel_size=[0.02] * 3, post_center_range=[float(v) for v in c[3:9]], max_num=int(c[2]),
                             score_threshold=None if c[1] < 0 else float(c[1]), alpha=float(c[0]), num_classes=10)
        res = coder.decode(preds)
        for b, r in enumerate(res):
            assert r["labels"].numpy().tolist() == z[f"s{si}_b{b}_labels"].tolist(), (si, b)
            for k in ("bboxes", "scores", "ious"):
                ref = z[f"s{si}_b{b}_{k}"]
                assert r[k].shape == ref.shape and np.abs(r[k].numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (si, b, k)
        si += 1
    assert si == 3


@pytest.mark.parametrize("spconv2", [False, True])
def test_checkpoint_round_trip_incl_spconv2_layout(tmp_path, spconv2):
    """save -> load (ref: extra_tools/test.py:197) restores every tensor bit for bit, also from a checkpoint whose sparse-conv weights
    are in the spconv 2.x layout [Cout,kD,kH,kW,Cin] and whose keys carry DDP's 'module.' prefix; a wrong shape is refused (identical
    detections from the reloaded model: tests/test_postproc_gpu.py)."""
    from uni3detr_amd.checkpoint import load_checkpoint, save_checkpoint
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    torch.manual_seed(1)
    a = build_model(copy.deepcopy(MODEL_CFG))
    path = str(tmp_path / "ck.pth")
    ck = save_checkpoint(a, path, meta=dict(epoch=3), to_spconv2=spconv2)
    if spconv2:
        w = ck["state_dict"]["pts_middle_encoder.conv_input.0.weight"]
        assert w.shape[0] == a.state_dict()["pts_middle_encoder.conv_input.0.weight"].shape[-1]      # Cout first
        ck["state_dict"] = {"module." + k: v for k, v in ck["state_dict"].items()}
        torch.save(ck, path)
    torch.manual_seed(2)
    b = build_model(copy.deepcopy(MODEL_CFG))
    out = load_checkpoint(b, path)
    assert out["meta"]["epoch"] == 3
    assert bool(out["meta"]["converted_sparse_weights"]) == spconv2
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    bad = {k: v for k, v in sa.items()}
    bad["pts_bbox_head.tgt_embed.weight"] = torch.zeros(5, 5)
    with pytest.raises(RuntimeError, match="shape mismatch"):
        load_checkpoint(b, dict(state_dict=bad))


class _Evil:
    """An object whose unpickling would run code (what a hostile checkpoint looks like)."""
    def __reduce__(self):
        return (os.path.join, ("a", "b"))


def test_checkpoint_with_python_objects_needs_explicit_trust(tmp_path, monkeypatch):
    """The tensors-only unpickler is the default; a file it rejects is loaded with the full (code-executing) unpickler ONLY on an
    explicit opt-in, and a missing file is an I/O error, not a reason to try the unsafe path."""
    from uni3detr_amd.checkpoint import load_checkpoint
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    m = build_model(copy.deepcopy(MODEL_CFG))
    path = str(tmp_path / "evil.pth")
    torch.save({"meta": {"obj": _Evil()}, "state_dict": {k: v for k, v in m.state_dict().items()}}, path)
    monkeypatch.delenv("U3D_TRUST_CHECKPOINTS", raising=False)
    with pytest.raises(RuntimeError, match="trusted=True"):
        load_checkpoint(m, path)
    assert load_checkpoint(m, path, trusted=True)["meta"]["obj"] == os.path.join("a", "b")
    with pytest.raises(FileNotFoundError):
        load_checkpoint(m, str(tmp_path / "absent.pth"))


def test_unselected_registry_names_match_reference_goldens():
    """RDIoULoss / RDIoUCost / SoftFocalLossCost / get_rdiou (no shipped config selects them) against vectors produced by the reference's
    own files (oracle/make_golden.py gen_extra); AxisAlignedIoU3DCost / RotatedIoU3DCost build from the registry."""
    import numpy as np
    import torch
    from uni3detr_amd.plugin import extra_costs as X
    from uni3detr_amd.registry import LOSSES, MATCH_COST
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "extra_costs.npz"))
    t = lambda k: torch.from_numpy(z[k])
    u, r = X.get_rdiou(t("p").unsqueeze(1), t("g").unsqueeze(0))
    assert torch.allclose(u, t("u"), rtol=1e-5, atol=1e-6) and torch.allclose(r, t("rdiou"), rtol=1e-5, atol=1e-7)
    cost = MATCH_COST.build(dict(type="RDIoUCost", weight=1.7))(t("p"), t("g"))
    assert torch.allclose(cost, t("rdiou_cost"), rtol=1e-5, atol=1e-6)
    sfc = MATCH_COST.build(dict(type="SoftFocalLossCost", weight=2.0))(t("cls"), t("labels"), t("iou"))
    assert torch.allclose(sfc, t("soft_focal_cost"), rtol=1e-5, atol=1e-6)
    a = t("a").requires_grad_(True)
    loss = LOSSES.build(dict(type="RDIoULoss", loss_weight=1.2))(a, t("b"), t("w"), avg_factor=5.0)
    assert abs(float(loss) - float(z["rdiou_loss"])) <= 1e-5 * abs(float(z["rdiou_loss"]))
    loss.backward()
    assert torch.allclose(a.grad, t("rdiou_loss_grad"), rtol=1e-4, atol=1e-6)
    # axis-aligned IoU cost: closed form on unit cubes shifted by half an edge -> IoU = 0.5 / 1.5
    aa = MATCH_COST.build(dict(type="AxisAlignedIoU3DCost", weight=2.0))
    b0 = torch.tensor([[0.0, 0, 0, 1, 1, 1]])
    b1 = torch.tensor([[0.5, 0, 0, 1.5, 1, 1], [2.0, 2, 2, 3, 3, 3]])
    assert torch.allclose(aa(b0, b1), torch.tensor([[-2.0 / 3.0, 0.0]]), atol=1e-6)
    assert MATCH_COST.build(dict(type="RotatedIoU3DCost", weight=1.0)).weight == 1.0


@pytest.mark.skipif(not os.path.isdir("/root/reference/projects/configs/uni3detr"), reason="reference tree not present")
def test_restated_pipelines_equal_the_shipped_config_files():
    import glob
    from uni3detr_amd.configs import pipelines as P
    from uni3detr_amd.registry import Config
    files = sorted(glob.glob("/root/reference/projects/configs/uni3detr/uni3detr_*.py"))
    assert len(files) == len(P.SHIPPED) == 6
    for f in files:
        name = os.path.basename(f)[len("uni3detr_"):-3]
        cfg = Config.fromfile(f)
        for key in ("train_pipeline", "test_pipeline"):
            ref, ours = getattr(cfg, key), P.SHIPPED[name][key]
            assert [e["type"] for e in ref] == [e["type"] for e in ours], (name, key)
            for r, o in zip(ref, ours):
                for a, v in o.items():
                    assert list(r[a]) == list(v) if isinstance(v, (list, tuple)) else r[a] == v, (name, key, a)
