"""GPU parity of the product path (HIP kernels behind the plugin classes) against
 (1) golden vectors produced by the reference's own head/transformer/assigner/loss files, and
 (2) the CPU oracle of the whole training forward (oracle/model.py) on a seeded synthetic scene.
fp32 mode; tolerance 1e-3 relative on logits / losses (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

import projects.mmdet3d_plugin  # noqa: F401
from oracle import model as om
from oracle.weights import seeded_input, seeded_tensor
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.plugin.structures import Boxes3D
from uni3detr_amd.registry import build_model
from uni3detr_amd.synth import room_scene

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "attn_drop"):
            m.attn_drop = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    return model


def seeded_model(cuda, seed, prefix_strip=""):
    model = build_model(MODEL_CFG)
    sd = {k: seeded_tensor(k[len(prefix_strip):] if k.startswith(prefix_strip) else k, tuple(v.shape), seed) for k, v in model.state_dict().items()}
    return model, sd


def test_head_forward_loss_backward_match_reference_golden(cuda):
    z = np.load(os.path.join(G, "head_train_b2.npz"))
    seed = int(z["seed"])
    model = build_model(MODEL_CFG)
    head = model.pts_bbox_head
    head.load_state_dict({k: seeded_tensor(k, tuple(v.shape), seed) for k, v in head.state_dict().items()})
    head = no_dropout(head).to(cuda).train()
    feats = seeded_input("pts_feats", (2, 256, 15, 40, 40), seed, -0.5, 1.0).clamp_min(0).to(cuda).requires_grad_(True)
    fps = seeded_input("fpsbpts", (2, 600, 3), seed, 0.0, 1.0).to(cuda)
    outs = head(feats, None, fps)
    for k, name in (("cls", "all_cls_scores"), ("box", "all_bbox_preds"), ("iou", "all_iou_preds")):
        ref = torch.from_numpy(z[k])
        err = (outs[name].detach().cpu() - ref).abs().max().item()
        assert err <= 1e-3 * max(1.0, ref.abs().max().item()), (k, err)
    gts, labels, o = [], [], 0
    for n in z["gt_lens"]:
        gts.append(Boxes3D(torch.from_numpy(z["gts"][o:o + n])).to(cuda))
        labels.append(torch.from_numpy(z["labels"][o:o + n]).to(cuda))
        o += n
    losses = head.loss(gts, labels, outs)
    assert np.array_equal(head._last_assigned.cpu().numpy().astype(np.int16), z["assigned"])
    for name, val in zip(z["loss_names"], z["loss_values"]):
        got = float(losses[str(name)])
        assert abs(got - val) <= 1e-3 * max(1.0, abs(val)), (name, got, val)
    sum(losses.values()).backward()
    g = feats.grad.reshape(-1)[::997].cpu().numpy()
    np.testing.assert_allclose(g, z["feats_grad_sub"], rtol=5e-3, atol=5e-5)
    pg = {k: float(p.grad.norm()) for k, p in head.named_parameters() if p.grad is not None}
    for k, ref in zip(z["pgrad_names"], z["pgrad_l2"]):
        assert abs(pg[str(k)] - ref) <= 2e-3 * max(ref, 1e-3) + 1e-5, (k, pg[str(k)], ref)


def test_head_eval_layout_matches_reference_golden(cuda):
    z = np.load(os.path.join(G, "head_eval_b1.npz"))
    seed = int(z["seed"])
    head = build_model(MODEL_CFG).pts_bbox_head
    head.load_state_dict({k: seeded_tensor(k, tuple(v.shape), seed) for k, v in head.state_dict().items()})
    head = head.to(cuda).eval()
    feats = seeded_input("pts_feats", (1, 256, 15, 40, 40), seed, -0.5, 1.0).clamp_min(0).to(cuda)
    fps = seeded_input("fpsbpts", (1, 600, 3), seed, 0.0, 1.0).to(cuda)
    with torch.no_grad():
        outs = head(feats, None, fps, rand_points=torch.from_numpy(z["rand_points"]).to(cuda))
    assert outs["all_cls_scores"].shape == (3, 1, 1200, 10)
    for k, name in (("cls", "all_cls_scores"), ("box", "all_bbox_preds"), ("iou", "all_iou_preds")):
        assert (outs[name].cpu() - torch.from_numpy(z[k])).abs().max().item() <= 1e-3 * max(1.0, float(np.abs(z[k]).max()))


@pytest.mark.parametrize("B", [2])
def test_full_training_forward_matches_cpu_oracle(cuda, B):
    seed = 11
    model = build_model(MODEL_CFG)
    sd = {k: seeded_tensor(k, tuple(v.shape), seed) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model = no_dropout(model).to(cuda).train()
    scenes = [room_scene(i, 20000 - 2500 * i) for i in range(B)]
    pts = [torch.from_numpy(s[0]) for s in scenes]
    gtb = []
    for s in scenes:
        g = torch.from_numpy(s[1]).clone()
        g[:, 2] -= g[:, 5] / 2
        gtb.append(g)
    labels = [torch.from_numpy(s[2]) for s in scenes]
    cfg = om.sunrgbd_cfg()
    with torch.no_grad():
        ref_losses, aux = om.forward_train(sd, [p.numpy() for p in pts], gtb, labels, cfg)
    feat, fpsb = model.extract_pts_feat([p.to(cuda) for p in pts])
    assert torch.equal(fpsb.cpu(), aux["fpsbpts"])                                   # FPS indices identical -> identical queries
    ref_feat = aux["feats"]
    err = (feat.detach().float().cpu() - ref_feat).abs().max().item()
    assert err <= 1e-3 * ref_feat.abs().max().item(), err
    losses = model.forward_pts_train(feat, [Boxes3D(g).to(cuda) for g in gtb], [l.to(cuda) for l in labels], None, None, fpsb)
    assert torch.equal(model.pts_bbox_head._last_assigned.cpu(), aux["assigned"])
    for k, v in ref_losses.items():
        assert abs(float(losses[k]) - float(v)) <= 1e-3 * max(1.0, abs(float(v))), (k, float(losses[k]), float(v))
    sum(losses.values()).backward()
    g = model.pts_middle_encoder.conv_input[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0
