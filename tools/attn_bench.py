"""Developer tool: attention forward / backward kernel times.  args: B H N [dropout_p]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import encoders, ops
dev = torch.device("cuda:0")
B, H, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 12, 197)
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
rng = ops.PhiloxState(dev, seed=1) if p > 0 else None
qkv = torch.randn(B, N, 3 * H * 64, device=dev).to(torch.bfloat16).requires_grad_()
cot = torch.randn(B, N, H * 64, device=dev).to(torch.bfloat16)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
with torch.no_grad():
    tf = timeit(lambda: encoders.self_attention(qkv, None, H, p, rng))
out = encoders.self_attention(qkv, None, H, p, rng)
def bw():
    qkv.grad = None
    out.backward(cot, retain_graph=True)
tb = timeit(bw)
fl = 4.0 * B * H * N * N * 64
print("B=%d H=%d N=%d p=%.2f: fwd %.1f us (%.0f TF useful)  bwd %.1f us (%.0f TF useful, 2.5x fwd flops)" % (B, H, N, p, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6))
