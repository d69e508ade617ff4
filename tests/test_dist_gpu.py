"""World size 2 on ONE GPU (two processes, gloo collectives on device tensors): the multi-rank flow of bench.py / TrainStep in graph
mode - capture first, process group afterwards, enable_dist() re-broadcast, per-step positive-count and flat-gradient all-reduces
between the graph replays - with different scenes per rank.  RCCL itself needs one GPU per rank (the driver's 2/4/8-GPU runs); what
this pins is everything around it: ranks stay bit-identical, the exchange really averages, the step trains."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, overlap, backend="gloo"):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import projects.mmdet3d_plugin  # noqa: F401
        from uni3detr_amd.configs.sunrgbd import model as MODEL_CFG
        from uni3detr_amd.plugin.structures import Boxes3D
        from uni3detr_amd.registry import build_model
        from uni3detr_amd.synth import room_scene
        from uni3detr_amd.trainer import TrainStep
        # gloo: both ranks share GPU 0 (collectives staged through the host); nccl (= RCCL): one GPU per rank, as the driver launches it
        dev = torch.device("cuda", rank if backend == "nccl" else 0)
        torch.cuda.set_device(dev)
        torch.manual_seed(100 + rank)                       # DIFFERENT initial weights per rank: enable_dist() must overwrite them
        model = build_model(copy.deepcopy(MODEL_CFG)).to(dev).train().set_precision("bf16")
        pts, gts, labels = [], [], []
        for i in range(2):                                  # two scenes per rank, different on every rank
            p, g, l = room_scene(10 * rank + i, 9000 + 1000 * rank)
            gb = torch.from_numpy(g).clone()
            gb[:, 2] -= gb[:, 5] / 2
            pts.append(torch.from_numpy(p).to(dev)); gts.append(Boxes3D(gb).to(dev)); labels.append(torch.from_numpy(l).to(dev))
        ts = TrainStep(model, pts, gts, labels, graph=True, lr=2e-4, overlap_reduce=overlap)
        assert ts.overlap == overlap
        snap = ts.snapshot()
        ts.capture()
        ts.restore(snap)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)      # after the captures, as bench.py does
        ts.enable_dist()
        assert ts.dist_on and ts.world == world
        p0 = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()])
        losses = [float(ts.step()) for _ in range(4)]
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()])
        g = ts.flat_grad.clone()
        # ranks agree bit for bit on parameters and on the exchanged gradient
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        gboth = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(gboth, g)
        starts = [torch.empty_like(p0) for _ in range(world)]
        dist.all_gather(starts, p0)
        ok = bool(torch.equal(both[0], both[1]) and torch.equal(gboth[0], gboth[1]) and torch.equal(starts[0], starts[1]))
        moved = float((flat - p0).abs().max())
        q.put((rank, ok, losses, moved, all(map(lambda v: v == v and abs(v) < 1e6, losses))))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                   # surface the failure in the parent
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc()[-1500:], 0.0, False))


@pytest.mark.parametrize("overlap", [False, True])
def test_world2_graph_step_keeps_ranks_identical(cuda, overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    out.sort()
    for rank, ok, losses, moved, finite in out:
        assert ok and finite, (rank, losses)
        assert moved > 0.0                                   # the optimizer stepped
    assert out[0][2] != out[1][2]                            # different scenes -> different local losses ...
    # ... but one shared model: rank 0's and rank 1's parameters were compared bit for bit inside the workers


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: runs on a >= 2-GPU node only")
def test_world2_rccl_graph_step_keeps_ranks_identical(cuda):
    """The same flow over RCCL (backend 'nccl'), one process per GPU - what `bench.py --gpus 2` does under torch.distributed.run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    env = dict(HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.update(env)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, True, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    out.sort()
    for rank, ok, losses, moved, finite in out:
        assert ok and finite, (rank, losses)
        assert moved > 0.0
    assert out[0][2] != out[1][2]
