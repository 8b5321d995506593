"""Developer tool: forward+backward of the unpruned CoAttention branch, for a kernel trace.  args: B N [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import fusion  # noqa: E402

B, N = int(sys.argv[1]), int(sys.argv[2])
dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.bfloat16
dev = torch.device("cuda:0")
D, NQ = 512, 30
torch.manual_seed(1)
m = fusion.CoAttention(embed_dim=D, num_heads=8, use_pruning=False).to(dev).to(dtype)  # as the step's fusion module
gen = torch.Generator(device=dev).manual_seed(3)
v = torch.randn(B, N, D, device=dev, generator=gen).to(dtype).requires_grad_()
q = torch.randn(B, NQ, D, device=dev, generator=gen).to(dtype).requires_grad_()
vg = torch.randn(B, D, device=dev, generator=gen).to(dtype)
qg = torch.randn(B, D, device=dev, generator=gen).to(dtype)
cot = torch.randn(B, 2 * D, device=dev, generator=gen)
for _ in range(13):
    v.grad = q.grad = None
    m.zero_grad(set_to_none=True)
    o = m(v, vg, q, qg)
    o.backward(cot.to(o.dtype))
torch.cuda.synchronize()
