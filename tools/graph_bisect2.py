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
torch.manual_seed(0)
model = build_model(MODEL_CFG).to(dev).train().set_precision("bf16")
data = make_batch(0, 8, 20000, dev)
ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=True)
ts.measure_capacities()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        if "warm1" in mode:
            ts._stage1()
        else:
            ts.eager_step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
if "drop" in mode:
    ts._outs = ts._T = ts._num_pos = ts._losses = None
    ts.loss = None
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
kw = dict(pool=torch.cuda.graph_pool_handle()) if "pool" in mode else {}
with torch.cuda.graph(g, stream=s, **kw):
    ts._stage1()
torch.cuda.synchronize()
print("captured stage1")
if "full" in mode:
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s, **kw):
        ts._stage2()
    torch.cuda.synchronize()
    print("captured stage2")
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, stream=s, **kw):
        ts._stage3()
    torch.cuda.synchronize()
    print("captured stage3")
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)
    for it in range(int(os.environ.get("NREP", "3"))):
        g.replay(); torch.cuda.synchronize(); print("r1", it, flush=True)
        g2.replay(); torch.cuda.synchronize(); print("r2", it, float(ts.loss), flush=True)
        print("   grad finite", bool(torch.isfinite(ts.flat_grad).all()), "norm", float(ts.flat_grad.norm()), flush=True)
        g3.replay(); torch.cuda.synchronize(); print("r3", it, flush=True)
        bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        print("   nonfinite params:", bad[:5], len(bad), flush=True)
        badb = [n for n, b in model.named_buffers() if b.dtype.is_floating_point and not torch.isfinite(b).all()]
        print("   nonfinite buffers:", badb[:5], len(badb), flush=True)
    print("loss", float(ts.loss))
print("MODE", mode, "OK")
