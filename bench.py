"""Benchmark of the Uni3DETR training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = voxelize -> SparseEncoderHD -> SECOND3D/FPN -> 2xFPS -> decoder/head -> device Hungarian -> losses ->
backward (RCCL gradient all-reduce overlapped) -> grad-clip -> AdamW, over a batch of B=8 synthetic SUN-RGB-D-shaped
scenes per GPU (20 000 points, 300 queries; BASELINE.json configs[1]).  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def make_batch(rank, B, npts, dev):
    from uni3detr_amd.plugin.structures import Boxes3D
    from uni3detr_amd.synth import room_scene
    pts, gts, labels = [], [], []
    for i in range(B):
        p, g, l = room_scene(rank * B + i, npts)
        gb = torch.from_numpy(g).clone()
        gb[:, 2] -= gb[:, 5] / 2
        pts.append(torch.from_numpy(p).to(dev))
        gts.append(Boxes3D(gb).to(dev))
        labels.append(torch.from_numpy(l).to(dev))
    return dict(points=pts, img_metas=None, gt_bboxes_3d=gts, gt_labels_3d=labels)


def cpu_baseline(npts):
    """The oracle restatement (pure torch CPU, fp32) timed fwd+bwd on ONE scene on this host's cores ("port")."""
    from oracle import model as om
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    from uni3detr_amd.synth import room_scene
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = build_model(MODEL_CFG)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "code_weights" not in k)
          for k, v in model.state_dict().items()}
    p, g, l = room_scene(0, npts)
    gb = torch.from_numpy(g).clone()
    gb[:, 2] -= gb[:, 5] / 2
    cfg = om.sunrgbd_cfg()
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        losses, _ = om.forward_train(sd, [p], [gb], [torch.from_numpy(l)], cfg)
        sum(losses.values()).backward()
        times.append(time.perf_counter() - t0)
        for v in sd.values():
            v.grad = None
    t = float(np.median(times[1:]))
    return dict(value=1.0 / t, unit="scenes/s", cores=cores, kind="port",
                sample=f"oracle/model.py fwd+bwd, fp32, 1 scene x {npts} pts, median of 2 after 1 warm-up ({t:.2f} s/scene)")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import projects.mmdet3d_plugin  # noqa: F401
    from uni3detr_amd import native as nv
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    nv.lib()                                            # fail loudly if the HIP library is absent

    torch.manual_seed(1234)
    model = build_model(MODEL_CFG).to(dev).train()      # constructor-default init == what the shipped flow trains from
    model.set_precision(args.precision)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-5, weight_decay=0.0001, fused=True)
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], gradient_as_bucket_view=True,
                                                        bucket_cap_mb=64, broadcast_buffers=False)
    data = make_batch(rank, args.batch, args.points, dev)

    def step():
        opt.zero_grad(set_to_none=True)
        losses = net(return_loss=True, **data)
        loss, _ = model._parse_losses(losses)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0, foreach=True)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    # ---- roofline census (untimed): pairs / algorithmic bytes of every sparse-conv launch of one step
    census = None
    if rank == 0 and not args.no_roofline:
        nv.TIMER = nv.KernelTimer("census")
        step()
        torch.cuda.synchronize()
        census = nv.TIMER.census
        nv.TIMER = None
    timer = None
    if rank == 0 and not args.no_roofline:
        timer = nv.TIMER = nv.KernelTimer("time")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    nv.TIMER = None
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    loss_val = float(last)

    if rank == 0:
        scenes = world * args.batch * args.steps
        out = {
            "metric": "scenes/sec (fwd+bwd) SUN-RGB-D 20k pts, 300 queries", "value": scenes / dt, "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "uni3detr_sunrgbd.py (BASELINE configs[1]): train step fwd+loss+bwd+clip+AdamW, "
                                   f"{args.batch} scenes/GPU x {args.points} pts, 300 queries x 3 groups, random-init weights",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "final_loss": loss_val},
        }
        if timer is not None and census:
            durs = timer.durations_ms()
            per_step = len(census)
            assert len(durs) == per_step * args.steps, (len(durs), per_step, args.steps)
            tot_ms = sum(d for _, d in durs)
            tot_bytes = sum(m["bytes"] for _, m in census) * args.steps
            tot_flops = sum(m["flops"] for _, m in census) * args.steps
            ach = tot_bytes / (tot_ms * 1e-3) / 1e9
            # the single heaviest launch of the step, by time
            per_call = np.array([d for _, d in durs]).reshape(args.steps, per_step).mean(0)
            j = int(per_call.argmax())
            mj = census[j][1]
            out["roofline"] = {
                "kernel": "k_spconv_fwd (SubMConv3d / SparseConv3d forward + dgrad, all launches of the step)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "launches_per_step": per_step, "ms_per_step": tot_ms / args.steps, "algorithmic_MB_per_step": tot_bytes / args.steps / 1e6,
                "tflops": tot_flops / (tot_ms * 1e-3) / 1e12,
                "heaviest_launch": {"tag": census[j][0], "ms": float(per_call[j]), "GBps": mj["bytes"] / (per_call[j] * 1e-3) / 1e9,
                                    "n_out": mj["n_out"], "cin": mj["cin"], "cout": mj["cout"], "pairs": mj["pairs"]},
            }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.points)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
