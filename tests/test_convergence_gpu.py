"""GPU: do the 16-bit modes TRAIN like the f32-grade one?  (bf16 training fidelity rested on an argument - per-tensor
gradient deviations of a random network - not on a convergence check.)  The same seeded weights, the same rotating batches at the
benched size (8 scenes x 20 000 points, 300 queries x 3 groups), the captured training step (fwd + match + loss + bwd + clip + AdamW), dropout
off, STEPS optimizer steps in each precision mode:
  * every mode's loss goes down (mean of the last 10 steps below the mean of the first 5 by a stated margin),
  * the bf16 and `mixed` trajectories stay within a stated band of the `parity` trajectory (window means), and no step is held.
What this pins is the claim DESIGN.md makes for the throughput mode: rounding noise in the encoder's gradients (40-90 % per tensor on
the random-init network) does not change where AdamW takes the loss over the first dozens of steps."""
import copy

import numpy as np
import pytest
import torch

import projects.mmdet3d_plugin  # noqa: F401
from oracle.weights import seeded_tensor
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.plugin.structures import Boxes3D
from uni3detr_amd.registry import build_model
from uni3detr_amd.synth import room_scene
from uni3detr_amd.trainer import TrainStep

pytestmark = pytest.mark.gpu
STEPS, B, NPTS, NBATCH = 60, 8, 20000, 4


def _batch(dev, index):
    pts, gts, labels = [], [], []
    for i in range(B):
        p, g, l = room_scene(index * B + i, NPTS)
        gb = torch.from_numpy(g).clone()
        gb[:, 2] -= gb[:, 5] / 2
        pts.append(torch.from_numpy(p).to(dev)); gts.append(Boxes3D(gb).to(dev)); labels.append(torch.from_numpy(l).to(dev))
    return pts, gts, labels


def _trajectory(dev, mode, batches):
    model = build_model(copy.deepcopy(MODEL_CFG))
    model.load_state_dict({k: seeded_tensor(k, tuple(v.shape), 31) for k, v in model.state_dict().items()})
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "attn_drop"):
            m.attn_drop = 0.0
    model = model.to(dev).train().set_precision(mode)
    ts = TrainStep(model, *batches[0], graph=True, lr=1e-4, overlap_reduce=True)
    snap = ts.snapshot()
    ts.capture(batches=batches)
    ts.restore(snap)
    packed = [(model.pack_points(p), model.pts_bbox_head.pack_gts(g, l, dev), None) for p, g, l in batches]
    losses = []
    for it in range(STEPS):
        ts.set_batch(*packed[it % len(packed)])
        losses.append(float(ts.step()))
    assert ts.held_steps() == 0 and all(np.isfinite(losses))
    return np.array(losses)


def test_all_precision_modes_follow_the_same_loss_trajectory(cuda):
    batches = [_batch(cuda, j) for j in range(NBATCH)]
    tr = {mode: _trajectory(cuda, mode, batches) for mode in ("parity", "mixed", "bf16")}
    win = lambda a, lo, hi: float(a[lo:hi].mean())
    print("\n[convergence] window means (steps 0-4 | 20-29 | last 10):",
          {m: (round(win(a, 0, 5), 3), round(win(a, 20, 30), 3), round(win(a, STEPS - 10, STEPS), 3)) for m, a in tr.items()})
    for mode, a in tr.items():
        assert win(a, STEPS - 10, STEPS) <= 0.93 * win(a, 0, 5), (mode, a[:5], a[-10:])           # the loss goes down in every mode
    ref = tr["parity"]
    for mode in ("mixed", "bf16"):
        a = tr[mode]
        assert abs(a[0] - ref[0]) <= 5e-3 * abs(ref[0]), (mode, a[0], ref[0])                      # same forward at step 0 (bf16 noise)
        for lo in range(0, STEPS, 10):
            d = abs(win(a, lo, lo + 10) - win(ref, lo, lo + 10)) / win(ref, lo, lo + 10)
            assert d <= 3e-2, (mode, lo, d, win(a, lo, lo + 10), win(ref, lo, lo + 10))          # measured: <= 0.5 % in every window (last 10: 7.413 | 7.430 | 7.412)
