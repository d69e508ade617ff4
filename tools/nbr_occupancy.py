"""How much of the sparse encoder's implicit-GEMM work multiplies rows that have no neighbour?  Per level of the bench workload:
share of (offset, row) pairs that exist, and share of (offset, 16-row block) / (offset, 128-row tile) pairs with at least one
existing pair (what a kernel that skips empty blocks would still have to do).  usage (GPU): python tools/nbr_occupancy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import projects.mmdet3d_plugin  # noqa: F401
    from uni3detr_amd import sparse as sp
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    dev = torch.device("cuda:0")
    model = build_model(MODEL_CFG).to(dev).train()
    data = bench.make_batch(0, 8, 20000, dev)
    pts = model.pack_points(data["points"])
    coors = model.voxelize_batch(pts)[0]
    enc = model.pts_middle_encoder
    lvl, _ = sp.level_from_coors(coors.int().contiguous(), 8, enc.sparse_shape)
    strides = [(2, 2, 2), (2, 2, 2), (2, 2, 2)]
    pads = [(1, 1, 1), (1, 1, 1), (0, 1, 1)]
    for i in range(4):
        fwd, _ = lvl.subm_tables()                        # [27, ld]
        n = int(lvl.n_dev.item())
        t = (fwd[:, :n] >= 0)
        pair = t.float().mean().item()
        def blocks(b):
            m = (n + b - 1) // b * b
            tt = torch.zeros(27, m, dtype=torch.bool, device=dev)
            tt[:, :n] = t
            return tt.view(27, m // b, b).any(-1).float().mean().item()
        print(f"level {i}: rows {n:7d}  existing (offset,row) pairs {pair:.3f}  non-empty 16-row blocks {blocks(16):.3f}  "
              f"32-row {blocks(32):.3f}  128-row tiles {blocks(128):.3f}", flush=True)
        if i < 3:
            lvl, _ = sp.strided_level(lvl, (3, 3, 3), strides[i], pads[i])


if __name__ == "__main__":
    main()
