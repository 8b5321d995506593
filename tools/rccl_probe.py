"""Single-rank RCCL probe: does nccl initialise on this box, does an all-reduce run eagerly, between
HIP-graph replays, and inside a captured graph?  (Developer tool; one process, world size 1.)"""
import os
import sys
import time
import faulthandler

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(90, exit=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(4_644_272, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print("eager all_reduce ok", float(x[0]), flush=True)
y = torch.zeros_like(x)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y.add_(x)
g.replay()
dist.all_reduce(y)
g.replay()
dist.all_reduce(y)
torch.cuda.synchronize()
print("graph replay + eager all_reduce ok", float(y[0]), flush=True)
try:
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        y.add_(x)
        dist.all_reduce(y)
    g2.replay()
    torch.cuda.synchronize()
    print("all_reduce inside a captured graph ok", float(y[0]), flush=True)
except Exception as exc:  # noqa: BLE001
    print("capture of all_reduce failed:", type(exc).__name__, str(exc)[:200], flush=True)
t0 = time.perf_counter()
for _ in range(20):
    dist.all_reduce(x)
torch.cuda.synchronize()
print(f"18.6 MB all_reduce (1 rank): {(time.perf_counter() - t0) / 20 * 1e6:.1f} us", flush=True)
dist.destroy_process_group()
