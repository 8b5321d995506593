"""Developer tool: cost of in-kernel noise on the gather / prune+mix kernels (B=512, N=197, D=512)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, N, D = 512, 197, 512
K = ops.keep_count(N, 0.5)
gen = torch.Generator(device=dev).manual_seed(1)
rng = ops.PhiloxState(dev, seed=3)
for dtype in (torch.bfloat16, torch.float32):
    seqs = [torch.randn(B, N, D, device=dev, generator=gen).to(dtype) for _ in range(6)]
    w = torch.rand(B, N, device=dev, generator=gen)
    idx, _ = ops.topk_indices(w, K, need_bottom=False)

    def timed(fn):
        for i in range(3):
            fn(seqs[i])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            fn(seqs[i % 6])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 30 * 1e3

    for kind in (None, "gaussian", "uniform"):
        t_g = timed(lambda s: ops.gather_mix(s, idx, 0.05, noise_type=kind, noise_scale=0.01, rng=rng))
        t_p = timed(lambda s: ops.prune_mix(s, w, K, 0.05, noise_type=kind, noise_scale=0.01, rng=rng))
        t_pool = timed(lambda s: ops.pruned_pool([dict(seq=s, idx=idx, alpha=0.05, noise_type=kind, noise_scale=0.01,
                                                       rng=rng)], torch.float32))
        print(f"{str(dtype)[6:]:9s} noise={str(kind):9s} gather_mix {t_g:7.1f} us  prune_mix {t_p:7.1f} us  "
              f"pruned_pool {t_pool:7.1f} us", flush=True)
