import torch
dev = torch.device("cuda:0")
s = torch.cuda.Stream()
pool = torch.cuda.graph_pool_handle()
out1 = torch.empty(1 << 20, device=dev); out2 = torch.empty(4096, 256, device=dev)
x = torch.randn(7200, 256, device=dev).bfloat16()
sums = torch.empty(256, device=dev)
def f1():
    a = torch.zeros(1 << 20, device=dev)       # memset node
    a += 1
    out1.copy_(a)
def f2():
    b = torch.zeros(4096, 256, device=dev)
    b.add_(2)
    out2.copy_(b)
    sums.copy_(x.float().sum(0))               # multi-block reduce (semaphores)
with torch.cuda.stream(s):
    for _ in range(2): f1(); f2()
torch.cuda.synchronize()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, pool=pool, stream=s): f1()
with torch.cuda.graph(g2, pool=pool, stream=s): f2()
ref = x.float().sum(0)
bad = 0
for it in range(200):
    g1.replay(); g2.replay()
    torch.cuda.synchronize()
    ok1 = bool((out1 == 1).all()); ok2 = bool((out2 == 2).all()); ok3 = bool(torch.isfinite(sums).all()) and float((sums - ref).abs().max()) < 1e-2
    if not (ok1 and ok2 and ok3):
        bad += 1
        if bad < 5: print("replay", it, ok1, ok2, ok3, float(out1.max()), float(out2.max()), float((sums - ref).abs().max()))
print("bad replays:", bad, "of 200")
