"""Benchmark of the Uni3DETR training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N>1: one rank per GPU - started by torch.distributed.run, or, when
                                                          run bare, by bench.py itself: uni3detr_amd/launch.py)

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
MFMA_PEAK_TF = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks (no 2:1 sparsity)


# BASELINE.json configs[1..4]: scenes per GPU and points per scene of the synthetic clouds (the room generator scaled to
# each configuration's range; nuScenes = 10 sweeps, 5 point features, 2 scenes per GPU)
WORKLOADS = {
    "sunrgbd": dict(batch=8, points=20000, file="uni3detr_sunrgbd.py", baseline="configs[1]"),
    "scannet_large": dict(batch=4, points=100000, file="uni3detr_scannet_large.py", baseline="configs[2]"),
    "kitti_3classes": dict(batch=4, points=18000, file="uni3detr_kitti_3classes.py", baseline="configs[3]"),
    "nuscenes": dict(batch=2, points=250000, file="uni3detr_nuscenes.py", baseline="configs[4]"),
}


def workload_cfg(name):
    import copy
    from uni3detr_amd.configs import variants
    return copy.deepcopy(getattr(variants, name))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batches", type=int, default=16, help="distinct synthetic batches rotated through the timed steps")
    ap.add_argument("--config", default="sunrgbd", choices=sorted(WORKLOADS),
                    help="shipped configuration to run (BASELINE.json configs[1..4]); the headline metric is quoted on sunrgbd")
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU (default: the workload's)")
    ap.add_argument("--points", type=int, default=None, help="points per scene (default: the workload's)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "mixed", "parity"],
                    help="bf16: BASELINE configs[1]; mixed: the reference's recipe (fp32 encoder + backbone, 16-bit neck + head); fp32: parity mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--allow-eager", action="store_true", help="if the hipGraph capture fails, fall back to eager launches instead of exiting non-zero")
    ap.add_argument("--no-modes", action="store_true", help="skip the `modes` block (mixed / parity throughput + logits deviation beside the bf16 headline)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` block (BASELINE.json configs[2..4] beside the headline)")
    ap.add_argument("--workload-steps", type=int, default=30, help="timed steps of each entry of the `workloads` block (5 warm-up steps, 4 rotating batches)")
    return ap.parse_args()


def make_batch(rank, B, npts, dev, index=0, cfg=None):
    from uni3detr_amd.plugin.structures import Boxes3D
    from uni3detr_amd.synth import SUNRGBD_RANGE, room_scene
    rng_range = tuple(cfg["pts_voxel_layer"]["point_cloud_range"]) if cfg is not None else SUNRGBD_RANGE
    nfeat = cfg["pts_middle_encoder"]["in_channels"] if cfg is not None else 4
    ncls = cfg["pts_bbox_head"]["num_classes"] if cfg is not None else 10
    pts, gts, labels = [], [], []
    for i in range(B):
        p, g, l = room_scene((index * 64 + rank) * B + i, npts, pc_range=rng_range)
        if nfeat > 4:                                    # nuScenes: (x, y, z, intensity, sweep time) - the two extras as zeros
            p = np.concatenate([p, np.zeros((p.shape[0], nfeat - 4), np.float32)], 1)
        l = l % ncls
        gb = torch.from_numpy(g).clone()
        gb[:, 2] -= gb[:, 5] / 2
        pts.append(torch.from_numpy(p).to(dev))
        gts.append(Boxes3D(gb).to(dev))
        labels.append(torch.from_numpy(l).to(dev))
    return dict(points=pts, img_metas=None, gt_bboxes_3d=gts, gt_labels_3d=labels)


def launch_class(tag, meta):
    """(arithmetic type of the kernel, direction): bf16 LDS-DMA / direct kernels vs the exact-f32 MFMA kernels (parity mode, and the
    encoder + backbone of `mixed`); forward and input gradient share a kernel, the weight gradient has its own."""
    return ("bf16" if meta.get("v2") else "f32", "wgrad" if "wgrad" in tag else "fwd")


def cpu_baseline(npts, budget_s=40.0, warmups=2, timed=5):
    """The oracle restatement (pure torch CPU, fp32) timed fwd+bwd on ONE scene on this host's cores ("port"), SURVEY.md 8(d)'s protocol:
    2 warm-up iterations, then the median of 5 timed ones.  Bounded: the loop stops early once `budget_s` seconds of CPU work are spent
    (at least one timed iteration is always measured; the sample string says what ran)."""
    from oracle import model as om
    from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
    from uni3detr_amd.registry import build_model
    from uni3detr_amd.synth import room_scene
    host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)   # cores this process may run on
    # threads actually used: this fp32 workload (many small sparse gathers + 256-channel 3-D convolutions on ONE scene) stops scaling
    # near 32 threads, and with one thread per core of a 256-core host it ran 350x SLOWER (1790 s vs 5 s per scene: oversubscribed
    # intra-op pools; HISTORY.md round 2) - `cores` reports the threads used, `host_cores` what the box offers
    cores = min(host_cores, 32)
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
    t_start = time.perf_counter()
    while len(times) < warmups + timed:
        t0 = time.perf_counter()
        losses, _ = om.forward_train(sd, [p], [gb], [torch.from_numpy(l)], cfg)
        sum(losses.values()).backward()
        times.append(time.perf_counter() - t0)
        for v in sd.values():
            v.grad = None
        if time.perf_counter() - t_start > budget_s and len(times) > 1:
            break
    n_warm = min(warmups, len(times) - 1)
    use = times[n_warm:]
    t = float(np.median(use))
    return dict(value=1.0 / t, unit="scenes/s", cores=cores, host_cores=host_cores, kind="port",
                thread_note="32 of the host's cores: the oracle's small gathers + one-scene 3-D convolutions stop scaling there; one thread per core of a 256-core host measured 350x slower (oversubscribed intra-op pools)",
                sample=f"oracle/model.py fwd+bwd, fp32, 1 scene x {npts} pts, {n_warm} warm-up + {len(use)} timed iteration(s), median ({t:.2f} s/scene)")


def time_mode(precision, args, dev, rot, cfg):
    """Throughput of ONE more precision mode on the benched workload, same protocol as the headline (capture over all rotating batches,
    `--warmup` untimed steps, `--steps` timed steps between synchronisations, rotating batches, loss checked finite).  N = 1 only."""
    from uni3detr_amd.registry import build_model
    from uni3detr_amd.trainer import TrainStep
    torch.manual_seed(1234)
    model = build_model(cfg).to(dev).train()
    model.set_precision(precision)
    d0 = rot[0]
    ts = TrainStep(model, d0["points"], d0["gt_bboxes_3d"], d0["gt_labels_3d"], graph=True, overlap_reduce=os.environ.get("U3D_OVERLAP_REDUCE", "1") == "1")
    snap = ts.snapshot()
    ts.capture(batches=[(d["points"], d["gt_bboxes_3d"], d["gt_labels_3d"]) for d in rot])
    ts.restore(snap)
    packed = [(model.pack_points(d["points"]), model.pts_bbox_head.pack_gts(d["gt_bboxes_3d"], d["gt_labels_3d"], dev), None) for d in rot]
    it = 0
    for _ in range(args.warmup):
        ts.set_batch(*packed[it % len(packed)]); it += 1
        ts.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        ts.set_batch(*packed[it % len(packed)]); it += 1
        last = ts.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ts.check_capacities()
    loss_val = float(last.detach())
    if not np.isfinite(loss_val):
        raise SystemExit(f"bench: {precision} training diverged (loss = {loss_val})")
    res = dict(value=args.batch * args.steps / dt, unit="scenes/s", ms_per_step=1000.0 * dt / args.steps, steps=args.steps, warmup=args.warmup,
               final_loss=loss_val, launch_mode="hipGraph", recaptures=int(ts.recaptures))
    del ts, model, packed
    torch.cuda.empty_cache()
    return res


def time_workload(name, args, dev, rank=0):
    """One more shipped configuration (BASELINE.json configs[2..4]) under the headline's protocol - bf16, hipGraph replay, capture over
    the rotating batches, untimed warm-up, timed steps between synchronisations, loss checked finite - at that configuration's own
    batch / cloud size.  N = 1, rank 0; fewer steps and rotating batches than the headline so the default run stays within minutes."""
    import types
    wl = WORKLOADS[name]
    cfg = workload_cfg(name)
    a = types.SimpleNamespace(batch=wl["batch"], points=wl["points"], steps=args.workload_steps, warmup=5)
    rot = [make_batch(rank, a.batch, a.points, dev, index=j, cfg=cfg) for j in range(4)]
    r = time_mode("bf16", a, dev, rot, cfg)
    r.update(workload=f"{wl['file']} (BASELINE {wl['baseline']})", batch=a.batch, points=a.points, precision="bf16")
    del rot
    torch.cuda.empty_cache()
    return r


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (ref: extra_tools/dist_train.sh:7-9): start the N ranks ourselves.
    Under torch.distributed.run (WORLD_SIZE set) this is skipped and the launcher's ranks are used as they are."""
    from uni3detr_amd.launch import LaunchError, spawn_ranks
    try:
        n_dev = torch.cuda.device_count()
        worst, codes = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], n_dev)
    except LaunchError as e:
        print(f"[bench] {e}", file=sys.stderr, flush=True)
        raise SystemExit(2)
    if worst:
        print(f"[bench] rank exit codes {codes}", file=sys.stderr, flush=True)
    raise SystemExit(worst)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    wl = WORKLOADS[args.config]
    args.batch = wl["batch"] if args.batch is None else args.batch
    args.points = wl["points"] if args.points is None else args.points
    import faulthandler
    # watchdog: a hung collective / kernel must not burn the whole GPU slot — dump all stacks and exit
    faulthandler.dump_traceback_later(int(os.environ.get("U3D_WATCHDOG_S", "900")), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench: rank {rank} wants GPU {local}, the node exposes {torch.cuda.device_count()}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    numa = None
    if world > 1 and os.environ.get("U3D_NUMA_BIND", "1") == "1":
        from uni3detr_amd.launch import bind_to_gpu_numa
        numa = bind_to_gpu_numa(local)                    # host threads of this rank next to its GPU (None: topology unknown)
    use_dist = world > 1 or os.environ.get("U3D_FORCE_DDP") == "1"      # the env flag exercises the RCCL path on one GPU

    def init_pg():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    if use_dist and args.no_graph:
        init_pg()

    import projects.mmdet3d_plugin  # noqa: F401
    from uni3detr_amd import native as nv
    from uni3detr_amd.registry import build_model
    MODEL_CFG = workload_cfg(args.config)
    nv.lib()                                            # fail loudly if the HIP library is absent

    torch.manual_seed(1234)
    model = build_model(MODEL_CFG).to(dev).train()      # constructor-default init == what the shipped flow trains from
    model.set_precision(args.precision)
    from uni3detr_amd.trainer import TrainStep
    data = make_batch(rank, args.batch, args.points, dev, cfg=MODEL_CFG)
    # the timed steps rotate through `--batches` distinct batches, resident in HBM and pre-packed; each step copies the next one into
    # the static input buffers (TrainStep.set_batch: device-to-device) - every sparse level sees changing row counts
    rot = [data] + [make_batch(rank, args.batch, args.points, dev, index=j, cfg=MODEL_CFG) for j in range(1, max(1, args.batches))]
    # two-phase backward: for N > 1 the flat-gradient all-reduce of the head / decoder / dense-stack slice rides under the encoder's
    # backward; the split itself is free (25.34 vs 25.44 ms on one GPU), so N = 1 runs the same schedule
    overlap = os.environ.get("U3D_OVERLAP_REDUCE", "1") == "1"
    # a sparse level outgrowing its captured capacity on ANY rank holds the step on all ranks (the flag rides the positive-count
    # all-reduce) and re-captures collectively: the process group is torn down for the capture and re-created afterwards
    pg_hooks = ((lambda: dist.destroy_process_group()), init_pg) if use_dist else None
    # U3D_GRAD_COMM=bf16: exchange the gradient in bf16 (half the xGMI bytes; the f32 exchange is the default and what the parity tests pin)
    comm_dt = torch.bfloat16 if os.environ.get("U3D_GRAD_COMM", "fp32") == "bf16" else torch.float32
    ts = TrainStep(model, data["points"], data["gt_bboxes_3d"], data["gt_labels_3d"], graph=not args.no_graph, overlap_reduce=overlap,
                   pg_hooks=pg_hooks, grad_comm_dtype=comm_dt)
    caps = None
    launch_mode = "eager" if args.no_graph else "hipGraph"
    census, marker, mark_targets = None, None, []
    if not args.no_graph:
        snap = ts.snapshot()
        if rank == 0 and not args.no_roofline and os.environ.get("U3D_GRAPH_MARKS", "1") == "1":
            # roofline timing INSIDE the replayed graphs: one exact-size eager step prices every conv launch (census); the heaviest
            # launches (equal flops: forward / input gradient / weight gradient of the 256-channel 3x3x3 layers) get an external
            # HIP event pair each, which the capture turns into event-record nodes around those kernels (native.KernelTimer 'mark')
            nv.TIMER = nv.KernelTimer("census")
            ts.eager_step()
            torch.cuda.synchronize()
            census = nv.TIMER.census
            # the heaviest launches (by flops) of EVERY kernel class - (bf16 | f32 kernel) x (forward / input gradient | weight gradient):
            # whichever class turns out to dominate the step's time is then priced from inside the replayed graph
            mark_targets = []
            for key in {launch_class(t, m) for t, m in census}:
                idx = [i for i, (t, m) in enumerate(census) if launch_class(t, m) == key]
                top = max(census[i][1]["flops"] for i in idx)
                mark_targets += [i for i in idx if census[i][1]["flops"] == top]
            marker = nv.TIMER = nv.KernelTimer("mark", mark_targets, per_step=len(census))
        try:
            # exact-size steps over every rotating batch -> capacities -> static-shape warm-up -> hipGraphs
            counts, caps = ts.capture(batches=[(d["points"], d["gt_bboxes_3d"], d["gt_labels_3d"]) for d in rot])
        except Exception as e:
            # a number measured on eager launches is NOT the headline configuration: without --allow-eager a failed capture fails the run
            if not args.allow_eager:
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); refusing to report an eager-mode number "
                      f"(--allow-eager to fall back)", file=sys.stderr, flush=True)
                raise SystemExit(3)
            nv.TIMER = marker = None
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches (--allow-eager)", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            ts._graphs = None
            ts.graph = False
            model.static_shapes = False
            model.pts_middle_encoder.level_capacities = None
            args.no_graph = True
            launch_mode = "eager (capture failed)"
        ts.restore(snap)                                # capture/warm-up iterations do not count as training
        if use_dist:
            init_pg()                                   # process group only AFTER the captures (see TrainStep.enable_dist)
    if use_dist:
        ts.enable_dist()
    packed = [(model.pack_points(d["points"]), model.pts_bbox_head.pack_gts(d["gt_bboxes_3d"], d["gt_labels_3d"], dev), None) for d in rot]
    it = [0]

    def step():
        ts.set_batch(*packed[it[0] % len(packed)])
        it[0] += 1
        return ts.step()

    for _ in range(args.warmup):
        step()
    ts.comm_diag = use_dist                               # event pairs around the gradient waits (three event records per step)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    comm = ts.comm_exposed_ms() if use_dist else None
    ts.comm_diag = False
    if not args.no_graph:
        ts.check_capacities()
    # ---- roofline: per-launch HIP-event timing needs individually launched kernels, so it runs on eager instrumented steps
    #      of the same workload right after the timed region (census pass prices each launch, timing pass measures it)
    nv.TIMER = None
    timer = None
    mark_ms = None
    MARK_REPLAYS = 10
    if marker is not None and not args.no_graph and marker.marks:
        # per-launch durations of the marked kernels as they run inside the replayed step: MARK_REPLAYS more replays (untimed),
        # events read after each
        ts.dist_on = False                               # rank 0 only from here on: no collectives
        acc = {i: [] for i in marker.marks}
        for _ in range(MARK_REPLAYS):
            step()
            for i, ms in marker.mark_durations_ms().items():
                acc[i].append(ms)
        mark_ms = {i: float(np.mean(v)) for i, v in acc.items() if v}
    ROOF_STEPS = 3
    if rank == 0 and not args.no_roofline:
        ts.dist_on = False                               # instrumented steps are local to rank 0: no collectives
        from uni3detr_amd.plugin import dense as _dense
        _dense.PARALLEL_BRANCHES = False                 # one stream: per-launch event timings must not overlap other work
        model.static_shapes = False                      # exact row counts: algorithmic bytes are priced on real sizes
        model.pts_middle_encoder.level_capacities = None
        if census is None:
            nv.TIMER = nv.KernelTimer("census")
            ts.eager_step()
            torch.cuda.synchronize()
            census = nv.TIMER.census
        timer = nv.TIMER = nv.KernelTimer("time")
        for _ in range(ROOF_STEPS):
            ts.eager_step()
        torch.cuda.synchronize()
        nv.TIMER = None
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    multi = None
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        # N > 1 diagnostics (the first multi-GPU run must explain itself): every rank's own wall time, its exposed gradient waits,
        # its re-captures and held steps, its NUMA placement
        mine = torch.tensor([dt_local * 1e3 / args.steps, (comm or {}).get("reduce_a_exposed_ms", 0.0), (comm or {}).get("reduce_b_exposed_ms", 0.0),
                             float(getattr(ts, "recaptures", 0)), float(ts.held_steps()), float(-1 if numa is None else numa["node"]),
                             (comm or {}).get("reduce_m_exposed_ms", 0.0)],
                            device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        rows = torch.stack(allr).cpu().numpy()
        multi = {"per_rank_ms_per_step": [round(float(v), 4) for v in rows[:, 0]],
                 "ms_per_step_min": float(rows[:, 0].min()), "ms_per_step_max": float(rows[:, 0].max()),
                 "reduce_a_exposed_ms": float(rows[:, 1].max()), "reduce_b_exposed_ms": float(rows[:, 2].max()),
                 "reduce_a_exposed_ms_per_rank": [round(float(v), 4) for v in rows[:, 1]],
                 "reduce_b_exposed_ms_per_rank": [round(float(v), 4) for v in rows[:, 2]],
                 "recaptures_per_rank": [int(v) for v in rows[:, 3]], "held_steps_per_rank": [int(v) for v in rows[:, 4]],
                 "numa_node_per_rank": [int(v) for v in rows[:, 5]],
                 # three buckets in reverse layer order: a = neck + head + decoder, in flight under SECOND3D's backward;
                 # m = SECOND3D, in flight under the sparse encoder's backward; b = the encoder, never overlapped (U3D_REDUCE_BUCKETS=2: a = a + m)
                 "reduce_m_exposed_ms": float(rows[:, 6].max()), "reduce_m_exposed_ms_per_rank": [round(float(v), 4) for v in rows[:, 6]],
                 "buckets": (comm or {}).get("buckets"),
                 "bucket_a_MB": (comm or {}).get("bucket_a_MB"), "bucket_m_MB": (comm or {}).get("bucket_m_MB"), "bucket_b_MB": (comm or {}).get("bucket_b_MB"),
                 "overlap_reduce": bool(ts.overlap),
                 "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None}
    dt = float(tt.item())
    loss_val = float(last.detach())
    if not np.isfinite(loss_val):
        raise SystemExit(f"bench: training diverged (loss = {loss_val}) — a throughput number for a broken step would be meaningless")

    joined = dist.get_world_size() if (use_dist and dist.is_initialized()) else 1          # ranks that actually joined the job
    if joined != world:
        raise SystemExit(f"bench: {joined} rank(s) joined the process group, {world} were launched")
    if rank == 0:
        nq_cfg = int(MODEL_CFG["pts_bbox_head"]["num_query"])
        scenes = joined * args.batch * args.steps
        out = {
            "metric": ("scenes/sec (fwd+bwd) SUN-RGB-D 20k pts, 300 queries" if args.config == "sunrgbd" else
                       f"scenes/sec (fwd+bwd) {args.config} {args.points} pts, {nq_cfg} queries"), "value": scenes / dt, "unit": "scenes/s",
            "n_gpus": joined, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp32": "f32", "mixed": "f32 encoder+backbone (convs as split-bf16: 3 bf16 MFMA products, f32 accumulation) / bf16 neck+head",
                      "parity": "f32 storage everywhere; convs as split-bf16 (3 bf16 MFMA products, f32 accumulation), decoder+head exact-f32 MFMA"}[args.precision], "data": "synthetic",
            "config": {"workload": f"{wl['file']} (BASELINE {wl['baseline']}): train step fwd+loss+bwd+clip+AdamW, "
                                   f"{args.batch} scenes/GPU x {args.points} pts, {nq_cfg} queries x 3 groups, random-init weights",
                       "global_batch": joined * args.batch, "parallelism": f"dp{joined}", "rccl_ranks": joined if use_dist else 0, "gradient_exchange_dtype": str(comm_dt).replace("torch.", ""),
                       "launcher": ("self (bench.py --gpus N)" if os.environ.get("U3D_SELF_LAUNCHED") == "1" else
                                    ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "single process")),
                       "final_loss": loss_val,
                       "launch_mode": launch_mode if launch_mode != "hipGraph" else (
                           ("hipGraph: " + ("voxelize | FPS on a second stream || encoder + dense stack | head + match" if isinstance((ts._graphs or [None])[0], tuple) else "fwd + match")
                            + (" | loss + bwd head/dense [all-reduce A overlaps] | bwd encoder | clip + AdamW" if ts.overlap else " | loss + bwd | clip + AdamW"))
                           + ", static-shape sparse levels"),
                       "sparse_level_capacities": caps, "rotating_batches": len(rot),
                       "fps_stream_calibration_ms": getattr(ts, "fps_stream_calibration_ms", None),
                       "fps_overlap_reference_ms": getattr(ts, "fps_overlap_reference_ms", None), "recaptures": int(getattr(ts, "recaptures", 0))},
        }
        if multi is not None:
            out["multi_gpu"] = multi
        if timer is not None and census:
            durs = timer.durations_ms()
            per_step = len(census)
            assert len(durs) == per_step * ROOF_STEPS, (len(durs), per_step, ROOF_STEPS)
            per_call = np.array([d for _, d in durs]).reshape(ROOF_STEPS, per_step).mean(0)          # ms, averaged over the instrumented steps
            calls = [dict(m, tag=t, ms=float(per_call[i])) for i, (t, m) in enumerate(census)]
            if os.environ.get("U3D_CONV_TABLE"):        # per-launch table of the conv launches of a step (shape, time): grouped, to stderr
                grp = {}
                for x in calls:
                    k = (x["kind"], x["tag"], x["n_in"], x["n_out"], x["cin"], x["cout"], x["kvol"])
                    g_ = grp.setdefault(k, [0, 0.0, 0.0])
                    g_[0] += 1; g_[1] += x["ms"]; g_[2] += x["flops"]
                for k, (cnt_, ms_, fl_) in sorted(grp.items(), key=lambda kv: -kv[1][1]):
                    print(f"[conv] {k[0]:6s} {k[1]:13s} n_in {k[2]:7d} n_out {k[3]:7d} {k[4]:4d}->{k[5]:4d} K={k[6]:2d}  x{cnt_:2d}  {ms_ * 1e3:8.1f} us/step  "
                          f"{ms_ / cnt_ * 1e3:7.1f} us each  {fl_ / ms_ / 1e9 if ms_ else 0:7.1f} TF/s", file=sys.stderr)

            def agg(sel):
                c = [x for x in calls if sel(x)]
                ms = sum(x["ms"] for x in c)
                by = sum(x["bytes"] for x in c)
                fl = sum(x["flops"] for x in c)
                return dict(launches_per_step=len(c), ms_per_step=ms, algorithmic_MB_per_step=by / 1e6,
                            GBps=(by / (ms * 1e-3) / 1e9) if ms else 0.0, TFLOPs=(fl / (ms * 1e-3) / 1e12) if ms else 0.0)

            # dominant kernel = the conv launch class with the most time; its heaviest single launch is priced.  With event nodes in
            # the graphs the launch duration is the one measured inside the replayed step (the same thing a rocprofv3 kernel trace of
            # the run shows); the per-class aggregates below stay on the eager instrumented steps (one event pair per launch)
            timing = "hip events around each launch, eager instrumented steps"
            if mark_ms:
                for i, ms in mark_ms.items():
                    calls[i]["ms_eager"] = calls[i]["ms"]
                    calls[i]["ms"] = ms
                # the marked launches are the heaviest shape of every kernel class (launch_class).  The dominant kernel is the class with
                # the most time per step over ALL of its launches; it is priced at the AVERAGE duration of its marked launches (the
                # contract's "average launch duration"); the slowest marked launch of that class is reported beside it
                tot = {}
                for x in calls:
                    k = launch_class(x["tag"], x)
                    tot[k] = tot.get(k, 0.0) + x["ms"]
                dom = max(tot, key=tot.get)
                mine = [i for i in mark_ms if launch_class(calls[i]["tag"], calls[i]) == dom] or list(mark_ms)
                j = max(mine, key=lambda i: mark_ms[i])
                dom_mean_ms = float(np.mean([mark_ms[i] for i in mine]))
                mark_ms = {i: mark_ms[i] for i in mine}
                timing = (f"hip event-record nodes inside the replayed hipGraph, mean over the {len(mine)} launches of this shape per step "
                          f"x {MARK_REPLAYS} replays")
            else:
                tot = {}
                for x in calls:
                    k = launch_class(x["tag"], x)
                    tot[k] = tot.get(k, 0.0) + x["ms"]
                dom = max(tot, key=tot.get)
                cand = [i for i, x in enumerate(calls) if launch_class(x["tag"], x) == dom]
                j = max(cand, key=lambda i: calls[i]["flops"])
                dom_mean_ms = None
            h = dict(calls[j])
            slowest_ms = max(mark_ms.values()) if mark_ms else h["ms"]
            if dom_mean_ms is not None:
                h["ms"] = dom_mean_ms
            ai = h["flops"] / h["bytes"]
            peak_tf = MFMA_PEAK_TF["bf16" if h.get("v2") else "f32"]          # the peak of the type THIS kernel computes in
            if ai > peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9):
                ach, peak, unit, bound = h["flops"] / (h["ms"] * 1e-3) / 1e12, peak_tf, "TFLOP/s", "mfma"
            else:
                ach, peak, unit, bound = h["bytes"] / (h["ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
            # HBM bytes per launch from the PMC passes of THIS build (profiles/pmc_traffic.json, tools/pmc_step.sh + pmc_to_json.py):
            # the entry must name the launched symbol at the launched grid and come from the same kernel source, else null
            is_wgrad = "wgrad" in h["tag"]
            sym = (("k_igemm_wgrad_glds8_256" if is_wgrad else "k_igemm_glds8_256x256") if h.get("v2") else
                   ("k_spconv_wgrad" if is_wgrad else "k_spconv_fwd"))
            traffic, traffic_note, pmc = None, None, None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                import hashlib
                pmc = json.load(open(tpath))
                srcf = os.path.join(ROOT, "uni3detr_amd", "csrc", "igemm_bf16.hip")
                cur = hashlib.sha256(open(srcf, "rb").read()).hexdigest()[:16] if os.path.exists(srcf) else None
                grid = (-(-h["n_out"] // 256)) * (-(-h["cout"] // 256)) * 512
                ent = pmc.get("kernels", {}).get(f"{sym}@{grid}") if not is_wgrad else None
                if is_wgrad:
                    cands = [v for k, v in pmc.get("kernels", {}).items() if k.startswith(sym + "@") and "hbm_read_bytes" in v]
                    ent = max(cands, key=lambda v: v["hbm_read_bytes"]) if cands else None
                if pmc.get("source_sha16", {}).get("igemm_bf16.hip") != cur:
                    traffic_note = "profiles/pmc_traffic.json was taken on a different igemm_bf16.hip: refused"
                elif ent is None or "hbm_read_bytes" not in ent or "hbm_write_bytes" not in ent:
                    traffic_note = f"no counter entry for {sym}"
                else:
                    traffic = ent["hbm_read_bytes"] + ent["hbm_write_bytes"]
                    traffic_note = f"{pmc.get('tag')}: {sym}, FETCH_SIZE x 2 + WRITE_SIZE per launch, L2 hit {ent.get('l2_hit_rate')}"
            out["roofline"] = {
                "kernel": sym + f" [{h['tag']}, {h['kind']} lattice, N={h['n_out']}, "
                          f"Cin={h['cin']}, Cout={h['cout']}, K={h['kvol']}, pairs={h['pairs']}]",
                "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_note,
                "launch_ms": h["ms"], "launch_timing": timing,
                "kernel_dtype": "bf16" if h.get("v2") else "f32", "time_per_step_by_kernel_class_ms": {f"{a}/{b}": round(v, 4) for (a, b), v in tot.items()},
                "frac_slowest_marked_launch": (h["flops"] / (slowest_ms * 1e-3) / 1e12 / peak if bound == "mfma" else None),
                "frac_mean_of_heaviest_launches": (float(np.mean([h["flops"] / (ms * 1e-3) / 1e12 for ms in mark_ms.values()])) / peak if mark_ms else None),
                "heaviest_launches_ms": ({calls[i]["tag"] + f"#{i}": round(ms, 4) for i, ms in sorted(mark_ms.items())} if mark_ms else None),
                "algorithmic_bytes": h["bytes"], "flops": h["flops"], "arithmetic_intensity": ai,
                "launch_GBps": h["bytes"] / (h["ms"] * 1e-3) / 1e9,
                "all_conv_launches": agg(lambda x: True),
                "submconv3d_sparse_fwd": agg(lambda x: x["kind"] == "sparse" and x["tag"] == "spconv_fwd" and x["kvol"] == 27 and x["n_in"] == x["n_out"]),
                "sparse_encoder_convs": agg(lambda x: x["kind"] == "sparse"),
                "dense_stack_convs": agg(lambda x: x["kind"] == "dense"),
                "hbm_peak_GBps": HBM_PEAK_GBS,
            }
            # the north_star's two named fractions, first-class: SubMConv3d forward vs the HBM roofline (algorithmic bytes / time of the
            # 27-offset submanifold launches, from this run's per-launch events) and the decoder attention kernels' MFMA utilisation
            # (counter-side: profiles/pmc_traffic.json of this build; null when no current counter pass exists)
            sub = out["roofline"]["submconv3d_sparse_fwd"]
            out["roofline"]["submconv3d_hbm_frac"] = sub["GBps"] / HBM_PEAK_GBS if sub["ms_per_step"] else None
            dec_src = None
            if pmc is not None:
                import hashlib
                ok = all(pmc.get("source_sha16", {}).get(f) == hashlib.sha256(open(os.path.join(ROOT, "uni3detr_amd", "csrc", f), "rb").read()).hexdigest()[:16]
                         for f in ("decoder.hip", "decoder_bwd.hip", "decoder_common.h"))
                att = {k.split("@")[0]: v.get("mfma_util") for k, v in pmc.get("kernels", {}).items() if k.startswith("k_mha_")}
                if ok and att:
                    out["roofline"]["decoder_attention_mfma_util"] = att
                    dec_src = pmc.get("tag")
                    rowk = {k.split("@")[0]: v.get("mfma_util") for k, v in pmc.get("kernels", {}).items() if k.startswith("k_dec_")}
                    out["roofline"]["decoder_row_kernels_mfma_util"] = rowk
            if dec_src is None:
                out["roofline"]["decoder_attention_mfma_util"] = None
            if pmc is not None:
                # EVERY kernel source the counter pass recorded is checked against the tree: a stale file is named, and nothing
                # derived from its kernels is quoted above (igemm_bf16.hip -> traffic, decoder*.hip -> attention utilisation)
                import hashlib
                stale = []
                for f, sha in sorted(pmc.get("source_sha16", {}).items()):
                    fp = os.path.join(ROOT, "uni3detr_amd", "csrc", f)
                    if not os.path.exists(fp) or hashlib.sha256(open(fp, "rb").read()).hexdigest()[:16] != sha:
                        stale.append(f)
                out["roofline"]["pmc_pass"] = {"tag": pmc.get("tag"), "sources_checked": sorted(pmc.get("source_sha16", {})), "stale_sources": stale}
            if os.environ.get("U3D_BENCH_DUMP_CALLS"):
                json.dump(calls, open(os.environ["U3D_BENCH_DUMP_CALLS"], "w"))
        if not args.no_cpu_baseline and world == 1 and args.config == "sunrgbd":      # the CPU leg is quoted on the headline workload only
            faulthandler.cancel_dump_traceback_later()       # CPU leg: no GPU work can hang here
            out["cpu_baseline"] = cpu_baseline(args.points)
            # checker leg: deviation of each mode from the fp32 CPU oracle on the same workload shape (2 scenes, seeded weights, dropout
            # off; the oracle's forward runs ONCE); gated by tests/test_bf16_parity_gpu.py (tolerances stated there)
            from oracle.parity_bf16 import bf16_deviation, oracle_reference
            oref = oracle_reference(B=2, npts=args.points)
            if args.precision in ("bf16", "mixed", "parity"):
                out[f"{args.precision}_vs_fp32_oracle"] = bf16_deviation(dev, B=2, npts=args.points, mode=args.precision, ref=oref)
            if args.precision == "bf16" and not args.no_modes and not args.no_graph:
                # the OTHER precision modes beside the headline: `mixed` = the reference's own recipe (fp32 encoder +
                # backbone, 16-bit neck + head), `parity` = f32-grade everywhere (the mode that holds logits within 1e-3) - throughput
                # under the headline's protocol and the measured deviation on the benched shape, both from THIS run
                out["modes"] = {}
                for mode in ("mixed", "parity"):
                    try:                                 # a failing extra mode must not cost the run its headline line: recorded, not raised
                        faulthandler.dump_traceback_later(int(os.environ.get("U3D_WATCHDOG_S", "900")), exit=True)
                        r = time_mode(mode, args, dev, rot, MODEL_CFG)
                        faulthandler.cancel_dump_traceback_later()
                        dv = bf16_deviation(dev, B=2, npts=args.points, mode=mode, ref=oref)
                        r["vs_fp32_oracle"] = dv
                        r["cls_logit_rel_l2"], r["box_rel_l2"], r["cls_logit_max_abs"] = dv["cls_logit_rel_l2"], dv["box_rel_l2"], dv["cls_logit_max_abs"]
                        r["logits_within_1e-3"] = bool(dv["cls_logit_rel_l2"] <= 1e-3 and dv["box_rel_l2"] <= 1e-3 and dv["iou_logit_rel_l2"] <= 1e-3)
                        out["modes"][mode] = r
                    except (Exception, SystemExit) as e:
                        faulthandler.cancel_dump_traceback_later()
                        print(f"[bench] modes.{mode} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                        out["modes"][mode] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.config == "sunrgbd" and args.precision == "bf16" and not args.no_workloads and not args.no_graph:
            # the other shipped configurations (ScanNet-large, KITTI, nuScenes were builder-run lines only): same
            # protocol, each at its own batch / cloud size; a failure is recorded, never raised
            out["workloads"] = {}
            for name in ("scannet_large", "kitti_3classes", "nuscenes"):
                try:
                    faulthandler.dump_traceback_later(int(os.environ.get("U3D_WATCHDOG_S", "900")), exit=True)
                    out["workloads"][name] = time_workload(name, args, dev, rank)
                    faulthandler.cancel_dump_traceback_later()
                except (Exception, SystemExit) as e:
                    faulthandler.cancel_dump_traceback_later()
                    print(f"[bench] workloads.{name} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                    out["workloads"][name] = {"error": f"{type(e).__name__}: {e}"}
        result_line = json.dumps(out)
    else:
        result_line = None
    if use_dist:
        dist.destroy_process_group()
    if result_line is not None:
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)      # RCCL prints banner lines through C stdio: flush them BEFORE the result line
        except Exception:
            pass
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
