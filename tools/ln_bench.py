"""Developer tool: LayerNorm(+residual) forward / backward kernel times at the ViT-B/16 batch-64 shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import encoders
dev = torch.device("cuda:0")
rows, D = 12608, 768
x = torch.randn(rows, D, device=dev).to(torch.bfloat16).requires_grad_()
r = torch.randn(rows, D, device=dev).to(torch.bfloat16).requires_grad_()
w = torch.ones(D, device=dev, dtype=torch.bfloat16, requires_grad=True); b = torch.zeros(D, device=dev, dtype=torch.bfloat16, requires_grad=True)
cy = torch.randn(rows, D, device=dev).to(torch.bfloat16); cs = torch.randn(rows, D, device=dev).to(torch.bfloat16)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
with torch.no_grad():
    tf = timeit(lambda: encoders.add_layer_norm(x, r, w, b, 1e-12))
y, s = encoders.add_layer_norm(x, r, w, b, 1e-12)
def bw():
    x.grad = r.grad = w.grad = b.grad = None
    torch.autograd.backward([y, s], [cy, cs], retain_graph=True)
tb = timeit(bw)
nb = rows * D * 2
print("LN+residual (%d x %d bf16): fwd %.1f us (%.2f TB/s of 4 passes)  bwd incl. param-grad fold %.1f us (%.2f TB/s of 4 passes)" % (rows, D, tf, 4 * nb / tf / 1e6, tb, 4 * nb / tb / 1e6))
