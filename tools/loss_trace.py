import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import projects.mmdet3d_plugin  # noqa
from bench import make_batch
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.registry import build_model
from uni3detr_amd.trainer import TrainStep
graph = sys.argv[1] == "graph"
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = build_model(MODEL_CFG).to(dev).train().set_precision("bf16")
data = make_batch(0, 8, 20000, dev)
if os.environ.get("NODROP") == "1":
    for m_ in model.modules():
        if isinstance(m_, torch.nn.Dropout): m_.p = 0.0
        if hasattr(m_, "attn_drop"): m_.attn_drop = 0.0
ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=graph)
if graph:
    snap = ts.snapshot(); ts.capture(); ts.restore(snap)
for i in range(16):
    l = ts.step()
    torch.cuda.synchronize()
    gn = float(ts.flat_grad.norm())
    if gn != gn:
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("   non-finite grads in", len(bad), "params; first:", bad[:6], flush=True)
        if i > 2: break
    print(i, float(l), "gradnorm", gn, {k: round(float(v), 3) for k, v in list(ts._losses.items())[:4]} if not graph else "", flush=True)
