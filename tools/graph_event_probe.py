"""Does hipEventRecordExternal inside a torch hipGraph capture give per-replay kernel timings on this ROCm?  (GPU)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni3detr_amd import native as nv

dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        c = a @ b
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
e0, e1, e2 = nv.ExternalEvent(), nv.ExternalEvent(), nv.ExternalEvent()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    d = a + 1
    e0.record()
    c = a @ b
    e1.record()
    c2 = c @ b
    e2.record()
    f = c2 * 2
torch.cuda.synchronize()
for i in range(4):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "mm1 ms", e0.elapsed_time(e1), "mm2 ms", e1.elapsed_time(e2), flush=True)
print("PROBE_OK")
