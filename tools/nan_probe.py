import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import projects.mmdet3d_plugin  # noqa
from bench import make_batch
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.registry import build_model
from uni3detr_amd.trainer import TrainStep
mode = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = build_model(MODEL_CFG).to(dev).train().set_precision("bf16")
B = int(os.environ.get("B", "8")); NP = int(os.environ.get("NP", "20000"))
data = make_batch(0, B, NP, dev)
ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=True)
if "fillzero" in mode:
    orig = ts._stage2
    def st2():
        ts.flat_grad.mul_(0.0)
        losses = ts.model.pts_bbox_head.loss_from_targets(ts._outs, ts._T, ts._num_pos)
        ts._losses = losses
        loss = sum(v for k, v in losses.items() if "loss" in k)
        loss.backward()
        ts.loss = loss.detach()
    ts._stage2 = st2
snap = ts.snapshot(); ts.capture(); ts.restore(snap)
g1, g2, g3 = ts._graphs
def finite(t): return bool(torch.isfinite(t).all())
for i in range(3):
    g1.replay(); torch.cuda.synchronize()
    print(i, "outs finite", {k: finite(v) for k, v in ts._outs.items()}, "T finite", {k: finite(v.float()) for k, v in ts._T.items()}, flush=True)
    g2.replay(); torch.cuda.synchronize()
    print(i, "loss", float(ts.loss), "flat_grad finite", finite(ts.flat_grad), "nan count", int((~torch.isfinite(ts.flat_grad)).sum()), flush=True)
    if not finite(ts.flat_grad):
        o = 0
        for n_, p_ in model.named_parameters():
            if not p_.requires_grad: continue
            g_ = ts.flat_grad[o:o + p_.numel()]
            if not finite(g_):
                idx = (~torch.isfinite(g_)).nonzero().flatten()[:6].tolist()
                print("   NaN grad in", n_, tuple(p_.shape), "idx", idx, "vals", g_[idx].tolist(), flush=True)
            o += p_.numel()
    if "nog3" not in mode:
        g3.replay(); torch.cuda.synchronize()
    print(i, "params finite", all(finite(p) for p in model.parameters()), flush=True)
