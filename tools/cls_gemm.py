"""Developer tool: the classifier's bf16 GEMMs alone (for counters / traces).
    python tools/cls_gemm.py <kind: fwd|dgrad|wgrad> [B] [out] [in]      (defaults: the 1024 -> 2048 layer at batch 1024)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops  # noqa: E402

kind = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
out_f = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
in_f = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(B, in_f, device=dev).bfloat16()
w = (torch.randn(out_f, in_f, device=dev) * 0.02).bfloat16()
b = torch.randn(out_f, device=dev).bfloat16()
gy = torch.randn(B, out_f, device=dev).bfloat16()
y = torch.empty(B, out_f, device=dev, dtype=torch.bfloat16)
gx = torch.empty(B, in_f, device=dev, dtype=torch.bfloat16)
gw = torch.empty(out_f, in_f, device=dev, dtype=torch.float32)
cs = torch.empty(out_f, device=dev, dtype=torch.float32)
fn = {"fwd": lambda: ops.gemm_bf16(x, False, w, False, B, out_f, in_f, y, bias=b),
      "dgrad": lambda: ops.gemm_bf16(gy, False, w, True, B, in_f, out_f, gx),
      "wgrad": lambda: ops.gemm_bf16(gy, True, x, True, out_f, in_f, B, gw, colsum=cs)}[kind]
for _ in range(10):
    fn()
torch.cuda.synchronize()
