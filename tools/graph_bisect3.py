import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import projects.mmdet3d_plugin  # noqa
from bench import make_batch
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.registry import build_model
from uni3detr_amd.trainer import TrainStep

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(MODEL_CFG).to(dev).train().set_precision("bf16")
data = make_batch(0, 8, 20000, dev)
ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=True)
ts.measure_capacities()
m = model
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        ts.eager_step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ts._outs = ts._T = ts._num_pos = ts._losses = None
ts.loss = None
torch.cuda.synchronize()
pool = torch.cuda.graph_pool_handle()
st = {}


def a():
    st["coors"], st["feats"], st["voxel_off"], st["cat"], st["scene_off"], st["lens"] = m.voxelize_batch(ts.pts)
def b():
    st["x0"] = m.pts_middle_encoder(st["feats"], st["coors"], 8)
def c():
    st["x"] = m.pts_neck(m.pts_backbone(st["x0"]))
def d():
    st["fps"] = m.fps_queries(st["cat"], st["scene_off"], st["lens"], st["coors"], st["voxel_off"])
def e():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ts._outs = m.pts_bbox_head(st["x"], None, st["fps"])
def f():
    ts._T = m.pts_bbox_head.loss_targets(ts.gts, None, ts._outs)
    ts._num_pos = ts._T["num_pos"].clone()


parts = [("vox", a), ("enc", b), ("dense", c), ("fps", d), ("head", e), ("targets", f), ("stage2", ts._stage2), ("stage3", ts._stage3)]
graphs = []
for name, fn in parts:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool, stream=s):
        fn()
    torch.cuda.synchronize()
    graphs.append((name, g))
    print("captured", name, flush=True)
faulthandler.dump_traceback_later(40, exit=True)
graphs = graphs[: int(os.environ.get("NPARTS", "8"))]
for it in range(3):
    for name, g in graphs:
        g.replay(); torch.cuda.synchronize(); print("replayed", it, name, flush=True)
    print("loss", float(ts.loss), flush=True)
print("ALL OK")
