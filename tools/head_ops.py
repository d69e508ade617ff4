"""Which ATen ops (count, shapes) the head forward, the targets and the loss launch in ONE eager step (GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import projects.mmdet3d_plugin  # noqa: F401
    from torch.profiler import ProfilerActivity, profile
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    from uni3detr_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = build_model(MODEL_CFG).to(dev).train()
    model.set_precision("bf16")
    data = bench.make_batch(0, 8, 20000, dev)
    ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=False)
    for _ in range(2):
        ts.eager_step()
    m = model
    with m.shadow_scope():
        feat, fps = m.extract_pts_feat(ts.pts)
        torch.cuda.synchronize()
        for seg in ("head_fwd", "targets", "loss", "backward_head_only"):
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
                if seg == "head_fwd":
                    with torch.autocast("cuda", dtype=m.amp_dtype):
                        outs = m.pts_bbox_head([f.detach().requires_grad_(True) for f in feat] if isinstance(feat, (list, tuple)) else feat.detach().requires_grad_(True), None, fps)
                elif seg == "targets":
                    T = m.pts_bbox_head.loss_targets(ts.gts, None, outs)
                    num_pos = T["num_pos"].clone()
                elif seg == "loss":
                    losses = m.pts_bbox_head.loss_from_targets(outs, T, num_pos)
                    loss = sum(v for k, v in losses.items() if "loss" in k)
                else:
                    loss.backward()
                torch.cuda.synchronize()
            ka = prof.key_averages(group_by_input_shape=True)
            rows = [(e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:80]) for e in ka if e.self_device_time_total > 0 and not e.key.startswith("void") and "Cijk" not in e.key and not e.key.startswith("k_")]
            nk = sum(e.count for e in ka if e.device_type == torch.autograd.DeviceType.CUDA) if hasattr(torch.autograd, "DeviceType") else 0
            tot = sum(r[2] for r in rows)
            print(f"==== {seg}: {sum(r[1] for r in rows)} op calls with device time, {tot / 1e3:.2f} ms self device time")
            for r in sorted(rows, key=lambda r: -r[2])[:28]:
                print(f"   {r[0][:44]:44s} x{r[1]:4d}  {r[2] / 1e3:7.3f} ms  {r[3]}")


if __name__ == "__main__":
    main()
