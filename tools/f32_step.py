"""Runs a few eager steps of the fp32 tier of bench.py's job (for rocprofv3 --kernel-trace; tools/trace_step.py
prints one steady-state step of the trace)."""
import sys

import torch

sys.path.insert(0, '/root/repo')
import bench  # noqa: E402

dev = torch.device('cuda:0')
fus, gsm, opt, main_opt, sets = bench.build_job(dev, 1024, torch.float32, 0)
step = bench.make_step(fus, gsm, opt, main_opt, sets, True)
for i in range(12):
    step(i)
torch.cuda.synchronize()
