import sys, torch
dev = torch.device("cuda:0")
variant = sys.argv[1]
s = torch.cuda.Stream()
xs = [torch.randn(7200, 256 if i % 3 else 512, device=dev).bfloat16() for i in range(60)]
refs = [x.float().sum(0) for x in xs]
ones = torch.ones(7200, device=dev, dtype=torch.bfloat16)
outs = [None] * 60
def red(x):
    if variant == "bf16sum": return x.sum(0)
    if variant == "f32sum": return x.float().sum(0)
    if variant == "sumdtype": return x.sum(0, dtype=torch.float32)
    if variant == "mv": return torch.mv(x.t(), ones)
    if variant == "matmul": return (ones[None] @ x)[0]
def f():
    for i, x in enumerate(xs):
        outs[i] = red(x)
with torch.cuda.stream(s):
    for _ in range(2): f()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s): f()
bad = 0; nan = 0; mx = 0.0
for it in range(50):
    g.replay(); torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        if not bool(torch.isfinite(o).all()): nan += 1; bad += 1; continue
        d = float((o.float() - r).abs().max()); mx = max(mx, d)
        if d > 2.0: bad += 1
print(variant, "bad:", bad, "nan:", nan, "of", 50 * 60, "max finite diff", mx)
