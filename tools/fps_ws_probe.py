"""Is the workspace tensor noted at capture time the buffer the replayed FPS graph writes?  (GPU; nuScenes workload)
Zero it between two replayed steps and look at what the next replay leaves in it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import projects.mmdet3d_plugin  # noqa: F401
    from uni3detr_amd import native as nv
    from uni3detr_amd.registry import build_model
    from uni3detr_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    cfg = bench.workload_cfg("nuscenes")
    torch.manual_seed(1234)
    model = build_model(cfg).to(dev).train()
    model.set_precision("bf16")
    data = bench.make_batch(0, 2, 250000, dev, cfg=cfg)
    ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=True)
    ts.capture(batches=[(data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"])])
    ws, off = nv._FPS_MULTI_WS
    print("noted workspace", tuple(ws.shape), hex(ws.data_ptr()), "flag offset", off)
    for it in range(3):
        ts.step()
    torch.cuda.synchronize()
    raw = ws.view(torch.uint8).reshape(-1)
    print("after 3 steps: int32 around the flag", raw[off - 32:off + 32].view(torch.int32).cpu().numpy())
    ws.zero_()
    torch.cuda.synchronize()
    ts.step()
    torch.cuda.synchronize()
    print("zeroed, one more step: slots head (u64)", raw[:32].view(torch.int64).cpu().numpy(), " around the flag", raw[off - 32:off + 32].view(torch.int32).cpu().numpy())
    print("nonzero bytes in the first 4 KiB:", int((raw[:4096] != 0).sum()), " in the rest:", int((raw[4096:] != 0).sum()))
    # does the captured memset node run on replay?  0xFF everywhere, one step: the slots of workgroups that do not exist (set 2 has 5
    # slices: slots 5..15 of both parities) must read zero afterwards, the bytes past the memset's 2112 stay 0xFF
    raw.fill_(255)
    torch.cuda.synchronize()
    ts.step()
    torch.cuda.synchronize()
    r = raw[:4096].cpu().numpy()
    s2 = 2 * 512
    print("set 2 parity 0 slots 5..15 (bytes", s2 + 80, "..", s2 + 256, "): distinct values", sorted(set(r[s2 + 80:s2 + 256].tolist()))[:8])
    print("bytes 2048..2112 (flag + pad):", r[2048:2112].tolist())
    print("bytes 2112..2176 (past the memset):", sorted(set(r[2112:2176].tolist()))[:8])
    # where exactly are the non-0xFF / nonzero runs?
    import numpy as np
    z = np.flatnonzero((r != 0) & (r != 255))
    print("first / last byte that is neither 0 nor 0xFF:", (int(z[0]), int(z[-1])) if z.size else None)


if __name__ == "__main__":
    main()
