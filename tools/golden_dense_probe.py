"""Diagnostic (GPU): per-quantity relative error of the product's dense stack against tests/golden/dense_stack.npz in every precision mode.
usage: python tools/golden_dense_probe.py [name]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projects.mmdet3d_plugin  # noqa: F401,E402
from tests.test_reference_golden_gpu import G, _cfg, _load, _rel  # noqa: E402
from uni3detr_amd import sparse as sp  # noqa: E402
from uni3detr_amd.registry import BACKBONES, NECKS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sunrgbd"
cuda = torch.device("cuda:0")
z = np.load(os.path.join(G, "dense_stack.npz"), allow_pickle=False)
seed = int(z["seed"])
for mode in ("fp32", "parity", "bf16"):
    m = _cfg(name)
    bb = _load(BACKBONES.build(m["pts_backbone"]), z[f"{name}.backbone_keys"], z[f"{name}.backbone_shapes"], seed).to(cuda).train()
    nk = _load(NECKS.build(m["pts_neck"]), z[f"{name}.neck_keys"], z[f"{name}.neck_shapes"], seed).to(cuda).train()
    x = torch.from_numpy(z[f"{name}.x"]).to(cuda).requires_grad_(True)
    amp = torch.bfloat16 if mode == "bf16" else None
    with sp.split_scope(mode == "parity"):
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            outs = bb(x)
            y = nk(outs)
        (y.float() * torch.from_numpy(z[f"{name}.cot"]).to(cuda)).sum().backward()
        torch.cuda.synchronize()
    named = {"pts_backbone." + k: p for k, p in bb.named_parameters()}
    named.update({"pts_neck." + k: p for k, p in nk.named_parameters()})
    print(f"== {name} {mode}")
    print("  fwd", [f"{_rel(o.float().cpu().detach().numpy(), z[f'{name}.backbone{i}']):.1e}" for i, o in enumerate(outs)],
          f"neck {_rel(y.float().cpu().detach().numpy(), z[f'{name}.neck']):.1e}")
    print(f"  dx {_rel(x.grad.float().cpu().numpy(), z[f'{name}.dx']):.1e}",
          f"wgrad_first {_rel(named['pts_backbone.blocks.0.0.weight'].grad[:8].float().cpu().numpy(), z[f'{name}.wgrad_first_8']):.1e}",
          f"wgrad_deconv2 {_rel(named['pts_neck.deblocks.2.0.weight'].grad[:, :4].float().cpu().numpy(), z[f'{name}.wgrad_deconv2_4']):.1e}")
    o = 0
    rows = []
    for k in z[f"{name}.bn_grad_keys"]:
        g = named[str(k)].grad.float().cpu().numpy().reshape(-1)
        ref = z[f"{name}.bn_grads"][o:o + g.size]
        o += g.size
        rows.append((str(k), _rel(g, ref)))
    print("  bn grads:", " ".join(f"{k.replace('pts_', '').replace('.weight', '.w').replace('.bias', '.b')}={e:.0e}" for k, e in rows))
