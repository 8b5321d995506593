"""Timing of the one-launch pruning branch (developer tool): vision only, text only, both."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, D, H = int(os.environ.get("B", 1024)), 512, 8
NV, NQ = int(os.environ.get("NV", 197)), 30
nbuf = 4
NOISE = os.environ.get("NOISE")  # gaussian | uniform | unset
ops.FUSED_ONE_PASS = os.environ.get("ONEPASS") is not None  # ONEPASS=1: the one-pass kernel
gen = torch.Generator(device=dev).manual_seed(1)
vs = [torch.randn(B, NV, D, device=dev, generator=gen).bfloat16() for _ in range(nbuf)]
qs = [torch.randn(B, NQ, D, device=dev, generator=gen).bfloat16() for _ in range(nbuf)]
w = torch.randn(3 * D, D, device=dev) / D ** 0.5
bias = torch.randn(3 * D, device=dev) * 0.1
qg = torch.randn(B, D, device=dev).bfloat16()
rng = ops.PhiloxState(dev, seed=3)


def spec(seq, N):
    k = ops.keep_count(N, 0.5)
    d = dict(q=qg, seq=seq, in_proj_weight=w, in_proj_bias=bias, num_heads=H, k=k, k_adv=ops.adv_count(k, 0.2),
             alpha=0.05)
    if NOISE and N == NV:  # the reference's default: noise on the vision tokens only (utils/fusion.py:211,221)
        d.update(noise_type=NOISE, noise_scale=0.01)
    return d


class P:
    def __init__(self):
        self.t = []
        self.o = None

    def __call__(self, name, phase):
        if name != "fused_prune_pool":
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if phase == "begin":
            self.o = e
        else:
            self.t.append((self.o, e))


for label, mk in (("vision", lambda i: [spec(vs[i % nbuf], NV)]), ("text", lambda i: [spec(qs[i % nbuf], NQ)]),
                  ("both", lambda i: [spec(vs[i % nbuf], NV), spec(qs[i % nbuf], NQ)])):
    p = P()
    ops.PROBE = p
    with torch.no_grad():
        for i in range(24):
            ops.fused_prune_pool(mk(i), torch.bfloat16, rng)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in p.t[4:])
    print(f"{label:7s} median {ts[len(ts)//2]*1e3:7.1f} us  min {ts[0]*1e3:7.1f} us", flush=True)
