"""GPU time of ONE eager training step by segment (encoder / head forward / targets / loss / backward / update), from torch.profiler
kernel records attributed to record_function ranges.  usage (GPU): python tools/seg_prof.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import projects.mmdet3d_plugin  # noqa: F401
    from torch.profiler import ProfilerActivity, profile, record_function
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    from uni3detr_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = build_model(MODEL_CFG).to(dev).train()
    model.set_precision("bf16")
    data = bench.make_batch(0, 8, 20000, dev)
    ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=False)
    for _ in range(3):
        ts.eager_step()
    torch.cuda.synchronize()
    m = model
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        with m.shadow_scope():
            with record_function("SEG_encoder"):
                feat, fps = m.extract_pts_feat(ts.pts)
                torch.cuda.synchronize()
            with record_function("SEG_head_fwd"):
                with torch.autocast("cuda", dtype=m.amp_dtype):
                    outs = m.pts_bbox_head(feat, None, fps)
                torch.cuda.synchronize()
        with record_function("SEG_targets"):
            T = m.pts_bbox_head.loss_targets(ts.gts, None, outs)
            num_pos = T["num_pos"].clone()
            torch.cuda.synchronize()
        with record_function("SEG_loss"):
            for p in ts.params:
                p.grad = None
            losses = m.pts_bbox_head.loss_from_targets(outs, T, num_pos)
            loss = sum(v for k, v in losses.items() if "loss" in k)
            torch.cuda.synchronize()
        with record_function("SEG_backward"):
            loss.backward()
            torch.cuda.synchronize()
    ev = prof.events()
    segs = [(e.name, e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("SEG_")]
    agg = {}
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CUDA or (hasattr(e, "device_type") and str(e.device_type).endswith("CUDA")):
            for name, a, b in segs:
                # kernels are stamped with GPU time; segments end with a synchronize, so launch-time containment is a fair attribution
                if a <= e.time_range.start <= b:
                    d = agg.setdefault(name, [0.0, 0])
                    d[0] += e.time_range.elapsed_us()
                    d[1] += 1
                    break
    for name, a, b in segs:
        d = agg.get(name, [0.0, 0])
        print(f"{name:14s} wall {1e-3 * (b - a):8.2f} ms   kernels {d[1]:5d}   kernel time {1e-3 * d[0]:8.2f} ms", flush=True)
    # backward split: kernels by name class inside SEG_backward
    for name, a, b in segs:
        if name != "SEG_backward":
            continue
        cls = {}
        for e in ev:
            if str(e.device_type).endswith("CUDA") and a <= e.time_range.start <= b:
                n = e.name
                k = ("conv" if ("igemm" in n or "spconv" in n or "tap_gather" in n) else "bn" if ("k_bn" in n or "col_stats" in n) else
                     "gemm" if n.startswith("Cijk") else "other-hip" if n.startswith("k_") or "k_" in n[:12] else "torch")
                d = cls.setdefault(k, [0.0, 0]); d[0] += e.time_range.elapsed_us(); d[1] += 1
        for k, d in sorted(cls.items(), key=lambda kv: -kv[1][0]):
            print(f"   backward/{k:10s} kernels {d[1]:5d}  time {1e-3 * d[0]:8.2f} ms")


if __name__ == "__main__":
    main()
