"""Which part of stage 1 breaks hipGraph capture?  python tools/graph_bisect.py <stage>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import projects.mmdet3d_plugin  # noqa
from bench import make_batch
from uni3detr_amd import native as nv
from uni3detr_amd import sparse as sp
from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
from uni3detr_amd.registry import build_model
from uni3detr_amd.trainer import TrainStep

stage = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(MODEL_CFG).to(dev).train().set_precision("bf16")
data = make_batch(0, 8, 20000, dev)
ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=True)
ts.measure_capacities()
m = model


def fn():
    if stage == "vox":
        return m.voxelize_batch(ts.pts)[0]
    coors, feats, voxel_off, cat, scene_off, lens = m.voxelize_batch(ts.pts)
    if stage == "level0":
        lvl, rank = sp.level_from_coors(coors.int().contiguous(), 8, m.pts_middle_encoder.sparse_shape)
        return lvl.coords
    if stage == "subm":
        lvl, rank = sp.level_from_coors(coors.int().contiguous(), 8, m.pts_middle_encoder.sparse_shape)
        return sp.subm_geom(lvl).nbr_fwd
    if stage == "enc":
        return m.pts_middle_encoder(feats, coors, 8)
    x = m.pts_middle_encoder(feats, coors, 8)
    if stage == "backbone":
        return m.pts_backbone(x)
    x = m.pts_neck(m.pts_backbone(x))
    if stage == "neck":
        return x
    fps = m.fps_queries(cat, scene_off, lens, coors, voxel_off)
    if stage == "fps":
        return fps
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = m.pts_bbox_head(x, None, fps)
    if stage == "head":
        return outs
    return m.pts_bbox_head.loss_targets(ts.gts, None, outs)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        fn()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = fn()
torch.cuda.synchronize()
import faulthandler
faulthandler.dump_traceback_later(40, exit=True)
for it in range(6):
    g.replay()
    torch.cuda.synchronize()
    print("replay", it, flush=True)
print("STAGE", stage, "OK")
