"""CPU, world size 2, gloo: the N>1 specific host logic — the fused scalar all-reduces of the loss / log-vars and the DDP
wrapping bench.py uses (gradient averaging with bucket views) on a CPU-runnable stand-in with the detector's call signature."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import projects.mmdet3d_plugin  # noqa: F401
        from uni3detr_amd.plugin.detector import Uni3DETR
        from uni3detr_amd.plugin.head import reduce_mean_
        # (1) per-layer positive counts: one vector all-reduce == mmdet reduce_mean applied per layer
        npos = torch.tensor([48.0, 40.0, 8.0]) * (rank + 1)
        got = reduce_mean_(npos.clone())
        assert torch.allclose(got, torch.tensor([48.0, 40.0, 8.0]) * 1.5)
        # (2) _parse_losses: loss = sum of 'loss' keys (local), log vars = rank means, one message
        det = Uni3DETR.__new__(Uni3DETR)
        losses = {"loss_cls": torch.tensor(1.0 + rank), "loss_bbox": torch.tensor(2.0), "d0.loss_cls": torch.tensor(3.0 * (rank + 1)),
                  "acc": torch.tensor(10.0 * rank)}
        loss, logs = Uni3DETR._parse_losses(det, losses)
        assert abs(float(loss) - (1.0 + rank + 2.0 + 3.0 * (rank + 1))) < 1e-6            # local loss drives backward
        assert abs(float(logs["loss_cls"]) - 1.5) < 1e-6 and abs(float(logs["acc"]) - 5.0) < 1e-6
        assert abs(float(logs["loss"]) - (1.5 + 2.0 + 4.5)) < 1e-6
        # (3) DDP exactly as bench.py wraps the detector (kwargs-only forward returning a loss dict)
        class Toy(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.a = torch.nn.Linear(4, 4)
                self.b = torch.nn.Linear(4, 1)

            def forward(self, return_loss=True, points=None, **kw):
                return {"loss_x": self.b(torch.relu(self.a(points))).pow(2).mean()}
        torch.manual_seed(0)
        toy = Toy()
        ref = Toy()
        ref.load_state_dict(toy.state_dict())
        net = torch.nn.parallel.DistributedDataParallel(toy, gradient_as_bucket_view=True, bucket_cap_mb=64, broadcast_buffers=False)
        xs = [torch.arange(8.0).view(2, 4) * (r + 1) for r in range(world)]
        net(return_loss=True, points=xs[rank])["loss_x"].backward()
        exp = [torch.zeros_like(p) for p in ref.parameters()]
        for r in range(world):
            ref.zero_grad()
            ref(points=xs[r])["loss_x"].backward()
            for e, p in zip(exp, ref.parameters()):
                e += p.grad / world
        for e, p in zip(exp, toy.parameters()):
            assert torch.allclose(p.grad, e, atol=1e-6)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_scalar_reductions_and_ddp():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
