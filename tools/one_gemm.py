"""One encoder-shaped GEMM a few times (developer tool, for counter runs).
    python tools/one_gemm.py <kind: nt|dgrad|wgrad|lib> [tokens] [out] [in]"""
import sys

import torch
import torch.nn.functional as F

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops  # noqa: E402

kind = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12608
out_f = int(sys.argv[3]) if len(sys.argv) > 3 else 3072
in_f = int(sys.argv[4]) if len(sys.argv) > 4 else 768
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(T, in_f, device=dev).bfloat16()
w = (torch.randn(out_f, in_f, device=dev) * 0.02).bfloat16()
b = torch.randn(out_f, device=dev).bfloat16()
gy = torch.randn(T, out_f, device=dev).bfloat16()
fn = {"nt": lambda: ops.linear_nt(x, w, b), "dgrad": lambda: ops.linear_nt(gy, w, None, True), "wgrad": lambda: ops.wgrad_tn(gy, x), "lib": lambda: F.linear(x, w, b)}[kind]
for _ in range(5):
    fn()
torch.cuda.synchronize()
