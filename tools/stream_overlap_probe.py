"""Do two streams overlap on this stack?  A 16-workgroup, ~1 ms kernel (k_fps: 16 point sets) on a side stream next to a chain of
chip-filling GEMMs on the default stream: wall time of both together vs each alone, for several side streams (torch's pool hands out
streams that may share a hardware queue with the default stream) and for graph replays of the same work."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni3detr_amd import native as nv

dev = torch.device("cuda:0")
torch.manual_seed(0)
nsets, n, m = 16, 20000, 300
base = torch.randn(nsets * n * 3, device=dev)
set_off = (torch.arange(nsets, device=dev) * n * 3).long()
set_n = torch.full((nsets,), n, dtype=torch.int32, device=dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def fps():
    return nv.fps(base, set_off, set_n, n, m)


def gemms(k=4):
    c = a
    for _ in range(k):
        c = (c @ b)
    return c


def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


t_f, t_g = wall(fps), wall(gemms)
print(f"alone: fps {t_f:.3f} ms, gemm chain {t_g:.3f} ms, serial sum {t_f + t_g:.3f} ms")
cur = torch.cuda.current_stream()
for i in range(10):
    s = torch.cuda.Stream(priority=-1 if i >= 6 else 0)

    def both():
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            fps()
        gemms()
        cur.wait_stream(s)
    print(f"side stream #{i} (id {s.stream_id:#x}, priority {s.priority}): both {wall(both):.3f} ms")

# graph replays on two streams
s = torch.cuda.Stream()
gs = torch.cuda.Stream()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
gs.wait_stream(cur)
with torch.cuda.graph(g1, stream=gs):
    o1 = fps()
with torch.cuda.graph(g2, stream=gs):
    o2 = gemms()
torch.cuda.synchronize()


def both_graphs():
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        g1.replay()
    g2.replay()
    cur.wait_stream(s)


print(f"graphs: fps graph alone {wall(g1.replay):.3f}, gemm graph alone {wall(g2.replay):.3f}, both on two streams {wall(both_graphs):.3f} ms")
for i in range(6):
    s = torch.cuda.Stream(priority=-1 if i >= 3 else 0)
    print(f"   graph replay with side stream #{i} (priority {s.priority}): {wall(both_graphs):.3f} ms")
