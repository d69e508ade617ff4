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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        ts.eager_step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=50,
                                                              max_shapes_column_width=70))


if __name__ == "__main__":
    main()
