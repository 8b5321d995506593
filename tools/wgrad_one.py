"""Developer tool: ten launches of the bf16 weight-gradient GEMM (3072 x 768 over 12608 rows) for counter passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops
dev = torch.device("cuda:0")
T, out_f, in_f = 12608, int(sys.argv[1]) if len(sys.argv) > 1 else 3072, 768
dy = torch.randn(T, out_f, device=dev).to(torch.bfloat16)
x = torch.randn(T, in_f, device=dev).to(torch.bfloat16)
dw = torch.empty(out_f, in_f, device=dev, dtype=torch.bfloat16)
for _ in range(10):
    ops.gemm_bf16(dy, True, x, True, out_f, in_f, T, dw)
torch.cuda.synchronize()
