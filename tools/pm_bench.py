"""Times vqaf_prune_mix at the module-boundary benchmark shape (B=128, N=577, D=512 bf16)."""
import sys
import torch
from vqa_amd import ops

dev = torch.device("cuda:0")
B, N, D = 128, 577, 512
gen = torch.Generator(device=dev).manual_seed(7)
seqs = [torch.randn(B, N, D, device=dev, generator=gen).bfloat16() for _ in range(6)]
w = torch.rand(B, N, device=dev, generator=gen)
keeps = [float(x) for x in sys.argv[1:]] or [0.25, 0.5, 0.75, 0.9]
for keep in keeps:
    K = ops.keep_count(N, 1 - keep)
    for k_adv in (0, ops.adv_count(K, 0.2) if N - K >= ops.adv_count(K, 0.2) else 0):
        rng = ops.PhiloxState(dev, seed=1)
        for _ in range(3):
            ops.prune_mix(seqs[0], w, K, 0.05, k_adv=k_adv, rng=rng)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for r in range(reps):
            ops.prune_mix(seqs[r % 6], w, K, 0.05, k_adv=k_adv, rng=rng)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = B * (N * D * 2 + N * 4 + (K + 1) * D * 2 + K * 8)
        print(f"keep {keep} k_adv {k_adv}: {ms * 1e3:.1f} us  {nbytes / ms / 1e6:.0f} GB/s", flush=True)
