"""Developer tool: the dense CoAttention forward+backward captured as a HIP graph and replayed a few times, for a
kernel trace of the replay (tools/prof_tool.sh <tag> tools/dense_graph_prof.py B N).  The last replay is
bracketed by two marker launches (cos_ of a small tensor)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import fusion  # noqa: E402

B, N = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
dtype = torch.bfloat16
torch.manual_seed(1)
m = fusion.CoAttention(embed_dim=512, num_heads=8, use_pruning=False).to(dev).to(dtype)
gen = torch.Generator(device=dev).manual_seed(3)
v = torch.randn(B, N, 512, device=dev, generator=gen).to(dtype).requires_grad_()
q = torch.randn(B, 30, 512, device=dev, generator=gen).to(dtype).requires_grad_()
vg = torch.randn(B, 512, device=dev, generator=gen).to(dtype)
qg = torch.randn(B, 512, device=dev, generator=gen).to(dtype)
cot = torch.randn(B, 1024, device=dev, generator=gen).to(dtype)


def fb():
    v.grad = q.grad = None
    m.zero_grad(set_to_none=True)
    m(v, vg, q, qg).backward(cot)


for _ in range(3):
    fb()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    fb()
for _ in range(5):
    graph.replay()
marker = torch.zeros(12345, device=dev)
torch.cuda.synchronize()
marker.cos_()
graph.replay()
marker.cos_()
torch.cuda.synchronize()
