"""How much of a training step is host enqueue time?  python tools/host_time.py [bf16|fp32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import projects.mmdet3d_plugin  # noqa
from bench import make_batch
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.registry import build_model

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(MODEL_CFG).to(dev).train().set_precision(prec)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=2e-5, fused=True)
data = make_batch(0, 8, 20000, dev)


def step(parts=None):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True)
    feat, fps = model.extract_pts_feat(data["points"]); t.append(time.perf_counter())
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=prec == "bf16"):
        outs = model.pts_bbox_head(feat, None, fps)
    t.append(time.perf_counter())
    losses = model.pts_bbox_head.loss(data["gt_bboxes_3d"], data["gt_labels_3d"], outs); t.append(time.perf_counter())
    loss, _ = model._parse_losses(losses)
    loss.backward(); t.append(time.perf_counter())
    torch.nn.utils.clip_grad_norm_(params, 10.0, foreach=True)
    opt.step(); t.append(time.perf_counter())
    if parts is not None:
        parts.append([b - a for a, b in zip(t[:-1], t[1:])])


for _ in range(3):
    step()
torch.cuda.synchronize()
host, total, parts = [], [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(parts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
import numpy as np
print("host enqueue ms", np.median(host) * 1e3, "total ms", np.median(total) * 1e3)
print("host parts ms [extract, head, loss, backward, clip+opt]:", (np.median(np.array(parts), 0) * 1e3).round(2))
