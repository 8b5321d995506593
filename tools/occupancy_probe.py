"""Developer tool: time vqaf_linear_nt at 256 / 512 / 1024 / 2048 output tiles (N = 128, K = 768): equal times up to
1024 tiles mean four workgroups per CU run concurrently."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
K, N = 768, 128
w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
for tiles in (128, 256, 512, 768, 1024, 1280, 2048):
    x = torch.randn(tiles * 128, K, device=dev).bfloat16()
    for _ in range(3):
        ops.linear_nt(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        e0.record()
        ops.linear_nt(x, w)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{tiles} tiles: {ts[10]:.1f} us", flush=True)
