"""Host time of each graph replay call of the captured training step (does a hipGraphLaunch block?) - prints per-call microseconds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import projects.mmdet3d_plugin  # noqa
from uni3detr_amd.registry import build_model
from uni3detr_amd.trainer import TrainStep

dev = torch.device("cuda:0")
cfg = bench.workload_cfg("sunrgbd")
torch.manual_seed(1234)
model = build_model(cfg).to(dev).train().set_precision("bf16")
data = bench.make_batch(0, 8, 20000, dev, cfg=cfg)
ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=True, overlap_reduce=True)
ts.capture()
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
g1, g2, g2b, g3 = ts._graphs
names = ["g1a", "gfps", "g1b", "g1c", "g2", "g2b", "g3"]
for it in range(4):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    cur, side = torch.cuda.current_stream(), ts._fps_stream
    g1[0].replay(); t.append(time.perf_counter())
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g1[1].replay()
    t.append(time.perf_counter())
    g1[2].replay(); t.append(time.perf_counter())
    cur.wait_stream(side)
    g1[3].replay(); t.append(time.perf_counter())
    g2.replay(); t.append(time.perf_counter())
    g2b.replay(); t.append(time.perf_counter())
    g3.replay(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    print("host us per replay call:", {n: round((b - a) * 1e6) for n, a, b in zip(names, t, t[1:])}, "| sync", round((t[-1] - t[-2]) * 1e6), "| total", round((t[-1] - t[0]) * 1e6))
