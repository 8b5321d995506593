"""Do timing events recorded INSIDE a captured HIP graph (external events) report per-replay
kernel durations?  Compares with eager event timing of the same kernel."""
import torch

dev = torch.device("cuda:0")
x = torch.randn(64 << 20, device=dev)
y = torch.empty_like(x)


def body(e0=None, e1=None):
    x.mul_(1.0001)
    if e0 is not None:
        e0.record()
    y.copy_(x)
    if e1 is not None:
        e1.record()
    x.add_(1e-6)


body()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
body(a, b)
torch.cuda.synchronize()
print("eager copy ms", a.elapsed_time(b))
e0 = torch.cuda.Event(enable_timing=True, external=True)
e1 = torch.cuda.Event(enable_timing=True, external=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body(e0, e1)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("graph copy ms", e0.elapsed_time(e1))
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(20):
    g.replay()
t1.record()
torch.cuda.synchronize()
print("20 replays ms each", t0.elapsed_time(t1) / 20, "last in-graph copy ms", e0.elapsed_time(e1))
