"""GPU: the U3D_* kernel-selection toggles.  Every import-time switch that picks a different kernel or fusion for the training step is
flipped (one at a time, in process) on the same model, weights and batch: the step must still run, and its loss and flat gradient
must agree with the default configuration's to bf16 noise.  Catches a toggle that rots (its fall-back path is otherwise never run)
and documents that the switches select implementations, not semantics."""
import copy
import importlib

import pytest
import torch

import projects.mmdet3d_plugin  # noqa: F401
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.plugin.structures import Boxes3D
from uni3detr_amd.registry import build_model
from uni3detr_amd.synth import room_scene
from uni3detr_amd.trainer import TrainStep

pytestmark = pytest.mark.gpu

# (module, attribute, flipped value)
TOGGLES = [
    ("uni3detr_amd.sparse", "SUBM_HALO", False),
    ("uni3detr_amd.sparse", "HALO_WGRAD", False),
    ("uni3detr_amd.sparse", "HALO_128", False),
    ("uni3detr_amd.sparse", "REV_SUBM_TABLE", False),
    ("uni3detr_amd.sparse", "STRIDED_DGRAD_SPLIT", False),
    ("uni3detr_amd.sparse", "NMAJOR_FWD", False),
    ("uni3detr_amd.sparse", "FUSED_CONV_STATS", False),
    ("uni3detr_amd.sparse", "BN_GRAD_FUSION", True),
    ("uni3detr_amd.plugin.sparse_encoder", "RESIDUAL_FUSION", False),
    ("uni3detr_amd.plugin.dense", "FANOUT_FUSION", False),
    ("uni3detr_amd.plugin.dense", "FUSED_UPSAMPLE_ORDER", False),
    ("uni3detr_amd.plugin.dense", "FUSED_LEVEL_SUM", False),
    ("uni3detr_amd.plugin.fused_decoder", "ENABLED", False),
    ("uni3detr_amd.plugin.fused_decoder", "PK_SCATTER", True),
    ("uni3detr_amd.plugin.fused_decoder", "SHARED_DEFER", False),
    ("uni3detr_amd.plugin.fused_decoder", "FUSED_REFINE_DECODE", False),
    ("uni3detr_amd.plugin.detector", "FUSED_FPS_GLUE", False),
    ("uni3detr_amd.plugin.head", "FUSED_QUERY_EMBED", False),
    ("uni3detr_amd.plugin.head", "FUSED_LOSS_TARGETS", False),
    ("uni3detr_amd.plugin.head", "FUSED_BOX_DECODE", False),
    ("uni3detr_amd.plugin.head", "FUSED_DET_LOSS", False),
    ("uni3detr_amd.plugin.transformer", "FUSED_LN", False),
    ("uni3detr_amd.plugin.transformer", "FUSED_SINE_EMBED", False),
    ("uni3detr_amd.plugin.transformer", "SHARED_VALUE_GRAD", False),
    ("uni3detr_amd.plugin.transformer", "OWN_WGRAD", False),
    ("uni3detr_amd.plugin.transformer", "RELU_EPILOGUE", False),
    ("uni3detr_amd.plugin.transformer", "SKINNY_WGRAD", True),
    ("uni3detr_amd.native", "PERMUTE_TILED", False),
]
EXACT = {("uni3detr_amd.plugin.fused_decoder", "SHARED_DEFER"), ("uni3detr_amd.plugin.detector", "FUSED_FPS_GLUE"),
         ("uni3detr_amd.plugin.head", "FUSED_LOSS_TARGETS")}
_BASE = {}


def _run(dev):
    pts, gts, labels = [], [], []
    for i in range(2):
        p, g, l = room_scene(i, 12000)
        gb = torch.from_numpy(g).clone()
        gb[:, 2] -= gb[:, 5] / 2
        pts.append(torch.from_numpy(p).to(dev)); gts.append(Boxes3D(gb).to(dev)); labels.append(torch.from_numpy(l).to(dev))
    torch.manual_seed(5)
    m = build_model(copy.deepcopy(MODEL_CFG))
    for mod in m.modules():                       # dropout off: runs must be comparable
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "attn_drop"):
            mod.attn_drop = 0.0
    m = m.to(dev).train().set_precision("bf16")
    ts = TrainStep(m, pts, gts, labels, graph=False)
    ts._stage1()
    ts._reduce_num_pos()
    ts._stage2()
    torch.cuda.synchronize()
    flat = ts.flat_grad.detach().float().clone()
    names = {id(p): n for n, p in m.named_parameters()}
    _BASE["slices"] = {names[id(p)]: (o, p.numel()) for p, o in zip(ts.params, ts.offsets)}
    return float(ts.loss), flat


@pytest.mark.parametrize("module,attr,value", TOGGLES, ids=[f"{m.split('.')[-1]}.{a}={v}" for m, a, v in TOGGLES])
def test_toggle_selects_an_implementation_not_a_result(cuda, module, attr, value):
    if "base" not in _BASE:
        _BASE["base"] = _run(cuda)
    loss0, g0 = _BASE["base"]
    mod = importlib.import_module(module)
    old = getattr(mod, attr)
    assert old != value, f"{module}.{attr}: the matrix must flip the default"
    setattr(mod, attr, value)
    try:
        loss, g = _run(cuda)
    finally:
        setattr(mod, attr, old)
    assert abs(loss - loss0) <= 2e-2 * abs(loss0), (loss, loss0)
    cos = float((g * g0).sum() / (g.norm() * g0.norm()))
    assert cos > 0.97 and abs(float(g.norm() / g0.norm()) - 1.0) < 0.1, (cos, float(g.norm() / g0.norm()))
    if (module, attr) in EXACT:
        # glue fusions: the same arithmetic in fewer launches - forward bit-identical, gradients to f32 summation order
        assert loss == loss0, (loss, loss0)
        worst = ("", 0.0)
        for name, (o, n) in _BASE["slices"].items():
            if "pts_bbox_head" not in name:          # (the conv stack's backward runs f32 atomics in the volume-gradient scatter: order noise)
                continue
            a, b = g[o:o + n], g0[o:o + n]
            d = float((a - b).norm() / b.norm().clamp(min=1e-12))
            worst = max(worst, (name, d), key=lambda t: t[1])
        assert worst[1] < 1e-4, worst
