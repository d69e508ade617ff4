"""torch.profiler view of ONE eager training step: which ATen ops (with shapes) the small torch kernels come from.
usage (GPU): python tools/op_prof.py > gpurun_out/op_prof.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import projects.mmdet3d_plugin  # noqa: F401
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
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack="--stack" in sys.argv,
                 experimental_config=(torch._C._profiler._ExperimentalConfig(verbose=True) if "--stack" in sys.argv else None)) as prof:
        ts.eager_step()
        torch.cuda.synchronize()
    if "--stack" in sys.argv:
        # attribute every kernel-launching ATen op to the python line (forward) or the autograd node (backward) that issued it
        agg = {}
        for e in prof.events():
            t = getattr(e, "self_device_time_total", 0) or 0
            if t <= 0 or not e.name.startswith("aten::"):
                continue
            key = None
            for fr in (e.stack or []):
                if "uni3detr_amd" in fr or "projects/" in fr:
                    key = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-70:]
                    break
            if key is None:
                par = e.cpu_parent
                while par is not None:
                    if "Backward" in par.name or "evaluate_function" in par.name:
                        key = par.name.replace("autograd::engine::evaluate_function: ", "bwd:")
                        break
                    par = par.cpu_parent
            k = (key or "?", e.name)
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1; a[1] += t
        tot = sum(a[1] for a in agg.values())
        print(f"aten ops with device time: {sum(a[0] for a in agg.values())} launches, {tot:.0f} us")
        for (key, name), (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:110]:
            print(f"{t:8.0f} us {cnt:4d}x  {name:28s} {key}")
        return
    if "--sums" in sys.argv:                       # the gradient sums the autograd engine issues itself: shapes + the node being evaluated
        for e in prof.events():
            t = getattr(e, "self_device_time_total", 0) or 0
            if t <= 0 or e.name not in ("aten::add_", "aten::add"):
                continue
            chain, par = [], e.cpu_parent
            while par is not None and len(chain) < 4:
                chain.append(par.name.replace("autograd::engine::evaluate_function: ", "bwd:")[:60])
                par = par.cpu_parent
            print(f"{t:6.0f} us {e.name:12s} {str(e.input_shapes)[:70]:70s} <- {' <- '.join(chain)}")
        return
    if "--aten" in sys.argv:                       # only the ATen launches, grouped by input shapes
        rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and (e.self_device_time_total or 0) > 0]
        rows.sort(key=lambda e: -e.self_device_time_total)
        print(f"aten launches: {sum(e.count for e in rows)}, {sum(e.self_device_time_total for e in rows):.0f} us")
        for e in rows[:80]:
            print(f"{e.self_device_time_total:8.0f} us {e.count:4d}x  {e.key:26s} {str(e.input_shapes)[:150]}")
        return
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=50,
                                                              max_shapes_column_width=70))


if __name__ == "__main__":
    main()
