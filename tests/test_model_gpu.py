"""GPU parity of the product path (HIP kernels behind the plugin classes) against
 (1) golden vectors produced by the reference's own head/transformer/assigner/loss files, and
 (2) the CPU oracle of the whole training forward (oracle/model.py) on a seeded synthetic scene.
fp32 mode; tolerance 1e-3 relative on logits / losses (BASELINE.json north_star)."""
import copy
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


import contextlib


@contextlib.contextmanager
def no_vendor_gemm():
    """Fails the test if the code inside reaches a vendor GEMM / attention entry point (F.linear -> hipBLASLt, SDPA -> AOTriton, mm,
    addmm, bmm): the 1e-3 reference parity must be earned by the hand-written decoder kernels the benchmark runs."""
    import torch.nn.functional as F

    def boom(name):
        def f(*a, **k):
            raise AssertionError(f"vendor kernel entry point reached: {name}")
        return f
    saved = [(F, "linear", F.linear), (F, "scaled_dot_product_attention", F.scaled_dot_product_attention), (torch, "mm", torch.mm),
             (torch, "addmm", torch.addmm), (torch, "bmm", torch.bmm), (torch, "matmul", torch.matmul),
             (torch.Tensor, "__matmul__", torch.Tensor.__matmul__), (torch, "_addmm_activation", torch._addmm_activation)]
    for mod, name, _ in saved:
        setattr(mod, name, boom(name))
    try:
        yield
    finally:
        for mod, name, fn in saved:
            setattr(mod, name, fn)


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
    with no_vendor_gemm():            # the decoder / head run on the f32 instantiation of the fused HIP kernels, not on ATen
        outs = head(feats, None, fps)
    assert head.transformer.decoder._fused_et == torch.float32
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
    with no_vendor_gemm():
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
    with torch.no_grad(), no_vendor_gemm():
        outs = head(feats, None, fps, rand_points=torch.from_numpy(z["rand_points"]).to(cuda))
    assert head.transformer.decoder._fused_et == torch.float32
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


@pytest.mark.parametrize("code", [8, 10])
def test_fused_detection_loss_matches_torch_formulation(cuda, code):
    """u3d_det_loss_fwd/_bwd (in-kernel forward-mode derivatives) == the torch formulation of Uni3DETRHead.loss_from_targets:
    all 12 loss scalars and the gradients w.r.t. class logits, box codes and IoU logits, incl. background rows and a scene without GT."""
    from uni3detr_amd.plugin import head as H
    torch.manual_seed(code)
    m = build_model(copy.deepcopy(MODEL_CFG)).pts_bbox_head.to(cuda)
    if code == 10:
        m.code_weights = torch.nn.Parameter(torch.tensor([1.0] * 8 + [0.2, 0.2], device=cuda), requires_grad=False)
    L, B, Q, C = 3, 4, 60, m.num_classes
    tdim = code - 1
    cls = torch.randn(L, B, Q, C, device=cuda)
    box = torch.randn(L, B, Q, code, device=cuda) * 0.5
    iou = torch.randn(L, B, Q, 1, device=cuda)
    w = (torch.rand(L, B, Q, device=cuda) < 0.3).float()
    w[:, 3] = 0                                                              # one scene without any assigned GT
    tgt = torch.cat([torch.randn(L, B, Q, 3, device=cuda), torch.rand(L, B, Q, 3, device=cuda) + 0.2,
                     (torch.rand(L, B, Q, 1, device=cuda) - 0.5) * 6.0] + ([torch.randn(L, B, Q, 2, device=cuda)] if code == 10 else []), -1)
    tgt = tgt * w.unsqueeze(-1)
    # matched predictions near their targets so that the IoU terms are exercised
    from uni3detr_amd.plugin.bbox import normalize_bbox
    box = torch.where(w.unsqueeze(-1) > 0, normalize_bbox(tgt)[..., :code] + 0.15 * torch.randn_like(box), box)
    lab = torch.where(w > 0, torch.randint(0, C, (L, B, Q), device=cuda), torch.full((L, B, Q), C, device=cuda))
    T = dict(w=w, tgt=tgt, lab=lab, asg=None, num_pos=H.layer_sums(w))
    num_pos = T["num_pos"].clone() * 0.75 + 0.5
    res = {}
    for fused in (False, True):
        H.FUSED_DET_LOSS = fused
        try:
            a, b, c = cls.clone().requires_grad_(True), box.clone().requires_grad_(True), iou.clone().requires_grad_(True)
            out = m.loss_from_targets({"all_cls_scores": a, "all_bbox_preds": b, "all_iou_preds": c}, T, num_pos)
            coef = {k: 0.5 + 0.1 * i for i, k in enumerate(sorted(out))}    # distinct upstream gradients per loss scalar
            sum(out[k] * coef[k] for k in out).backward()
            res[fused] = ({k: float(v) for k, v in out.items()}, a.grad.clone(), b.grad.clone(), c.grad.clone())
        finally:
            H.FUSED_DET_LOSS = True
    ref, got = res[False], res[True]
    assert set(ref[0]) == set(got[0]) and len(ref[0]) == 4 * L
    for k in ref[0]:
        assert abs(ref[0][k] - got[0][k]) <= 2e-4 * max(1.0, abs(ref[0][k])), (k, ref[0][k], got[0][k])
    for r, g, name in zip(ref[1:], got[1:], ("dcls", "dbox", "diou")):
        assert (r - g).abs().max().item() <= 2e-4 * max(1e-3, r.abs().max().item()), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("code", [8, 10])
def test_fused_box_decode_matches_torch_formulation(cuda, code, dtype):
    """u3d_box_decode_fwd/_bwd == the torch formulation of uni3detr_head.py:475-490 (inverse_sigmoid of the reference point, sigmoid +
    point-cloud range on columns 0,1,4): values and the gradients w.r.t. the branch output and the reference point, incl. reference
    points on / beyond the clamp edges."""
    from uni3detr_amd.plugin.head import _BoxDecode
    from uni3detr_amd.plugin.transformer import inverse_sigmoid
    torch.manual_seed(code)
    pr = [-3.2, -0.2, -2.0, 3.2, 6.2, 0.56]
    n = 1000
    tmp = (torch.randn(4, n // 4, code, device=cuda) * 1.5).to(dtype)
    ref = torch.rand(4, n // 4, 3, device=cuda)
    ref.view(-1)[:8] = torch.tensor([0.0, 1.0, 1e-6, 1 - 1e-6, 0.5, 2e-5, 1 - 2e-5, 0.25], device=cuda)
    gout = torch.randn(4, n // 4, code, device=cuda)
    a, ra = tmp.clone().requires_grad_(True), ref.clone().requires_grad_(True)
    out = _BoxDecode.apply(a, ra, tuple(pr))
    out.backward(gout)
    b, rb = tmp.clone().requires_grad_(True), ref.clone().requires_grad_(True)
    t = b.float().unbind(-1)
    rf = inverse_sigmoid(rb).unbind(-1)
    exp = torch.stack([(t[0] + rf[0]).sigmoid() * (pr[3] - pr[0]) + pr[0], (t[1] + rf[1]).sigmoid() * (pr[4] - pr[1]) + pr[1], t[2], t[3],
                       (t[4] + rf[2]).sigmoid() * (pr[5] - pr[2]) + pr[2], *t[5:]], -1)
    exp.backward(gout)
    assert out.dtype == torch.float32 and (out - exp).abs().max().item() <= 1e-5 * 10
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert (a.grad.float() - b.grad.float()).abs().max().item() <= tol * max(1.0, b.grad.float().abs().max().item())
    interior = (ref > 1e-4) & (ref < 1 - 1e-4)                   # on the clamp edges the one-sided derivative is a convention
    d = (ra.grad - rb.grad).abs()
    assert (d[interior] <= 1e-4 * rb.grad.abs()[interior].clamp_min(1.0)).all()
    assert torch.isfinite(ra.grad).all()
