"""Benchmark of the DP-SGD-trained fusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one DP-SGD training step of the fusion + classifier path on token-level synthetic
COCO-QA-shaped inputs already resident in HBM (encoders are outside the hot path):
zero_grad -> CoAttention(pruning, keep 0.5, adversarial mixing) forward -> classifier forward ->
mean CE -> backward (classifier input gradients, fusion input gradients) -> per-sample norms,
clip (C = 1.0), clipped sums, Gaussian noise (sigma = 1.0), RCCL all-reduce, AdamW step.
At N = 1 this is BASELINE.json configs[2] (logical batch 1024) on one GPU.  For N > 1 the default
keeps the per-GPU batch at 1024 (weak scaling: logical batch 1024*N, `value` = all samples / time);
`--scaling strong` shards the literal logical batch 1024 instead, and an N > 1 run also reports that
sharded configuration under "strong_scaling_1024" so both readings of configs[2] are measured.

Rank 0 prints ONE JSON line: whole-job samples/s plus a `roofline` object for the prune+mix
kernel (timed live with stream events inside the timed region), a `cpu_baseline` object (the
oracle, i.e. the reference's op sequence, timed on the host cores on a bounded sample) and the
prune+mix GB/s sweep of configs[3].
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak; the measured copy ceiling is ~6.3 TB/s

LOGICAL_BATCH = 1024
N_VIS, N_TXT, D, C = 197, 30, 512, 430
KEEP = 0.5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=LOGICAL_BATCH,
                    help="per-GPU batch (weak scaling) or logical whole-job batch (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python (no HIP graph)")
    ap.add_argument("--graph-multi", action="store_true",
                    help="with N > 1 also replay the two compute phases as HIP graphs around the eager "
                         "RCCL all-reduce (default for N > 1 is eager launches: the graph + collective "
                         "combination could not be validated on a one-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64)
    return ap.parse_args()


class Probe:
    """Stream-event pairs around one named kernel launch (torch's current stream is the stream
    the C ABI launches on)."""

    def __init__(self, name):
        self.name, self.pairs, self.enabled, self._open = name, [], False, None

    def __call__(self, name, phase):
        if not self.enabled or name != self.name:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if phase == "begin":
            self._open = ev
        else:
            self.pairs.append((self._open, ev))

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / max(1, len(self.pairs))


def build_job(dev, local_b, dtype, rank, logical_b=LOGICAL_BATCH):
    from vqa_amd import dp, fusion
    torch.manual_seed(42)  # the reference's seed (train.py:19): identical parameters on all ranks
    fus = fusion.CoAttention(embed_dim=D, num_heads=8, use_pruning=True, prune_ratio=1 - KEEP,
                             mixup_alpha=0.05, adversarial_mix=True, adv_ratio=0.2,
                             noise_injection=False).to(dev)
    cls = nn.Sequential(nn.Linear(2 * D, 2048), nn.BatchNorm1d(2048), nn.ReLU(), nn.Dropout(0.3),
                        nn.Linear(2048, 1024), nn.BatchNorm1d(1024), nn.ReLU(), nn.Linear(1024, C))
    cls = dp.ModuleValidator.fix(cls).to(dev)
    if dtype == torch.bfloat16:
        fus = fus.to(torch.bfloat16)
    gsm = dp.GradSampleModule(cls)
    inner = torch.optim.AdamW(cls.parameters(), lr=1e-4, weight_decay=0.01, fused=True, capturable=True)
    opt = dp.DPOptimizer(inner, gsm, noise_multiplier=1.0, max_grad_norm=1.0,
                         expected_batch_size=logical_b, seed=1234)
    main_opt = torch.optim.AdamW(fus.parameters(), lr=1e-4, weight_decay=0.01, fused=True, capturable=True)
    # four rotating input sets: > 256 MiB of tokens in total, so no set survives in the
    # Infinity Cache from one use to the next
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    sets = []
    for _ in range(4):
        v = torch.randn(local_b, N_VIS, D, device=dev, generator=gen).to(dtype).requires_grad_()
        q = torch.randn(local_b, N_TXT, D, device=dev, generator=gen).to(dtype).requires_grad_()
        y = torch.randint(0, C, (local_b,), device=dev, generator=gen)
        sets.append((v, v.detach().float().mean(1).to(dtype), q, q.detach().float().mean(1).to(dtype), y))
    return fus, gsm, opt, main_opt, sets


def make_step(fus, gsm, opt, main_opt, sets):
    """The training step in the order of utils/src.py:66-106, split at the one collective:
    `local(i)` = zero_grad, forward, loss, backward, main step, per-rank DP work;
    `reduce()` = the gradient all-reduce (RCCL; nothing at N = 1); `apply()` = the DP update."""
    from vqa_amd import ops

    def local(i):
        v, vg, q, qg, y = sets[i % len(sets)]
        v.grad = None
        q.grad = None
        main_opt.zero_grad()
        opt.zero_grad()
        fused = fus(v, vg, q, qg)
        logits = gsm(fused)
        loss = ops.cross_entropy_mean(logits, y)
        loss.backward()
        main_opt.step()
        opt.local_step()
        return loss

    def step(i):
        local(i)
        opt.reduce_step()
        opt.apply_step()

    step.local, step.reduce, step.apply = local, opt.reduce_step, opt.apply_step
    return step


def prune_mix_sweep(dev, dtype):
    """configs[3]: N = 577 vision tokens, B = 128, keep-ratio sweep; the module-boundary op
    PrivacyAwareTokenPruning.forward (top-k + gather + mix token) with its algorithmic bytes
    N*D*s + N*4 + (K+1)*D*s + K*8 per sample."""
    from vqa_amd import ops
    B, N = 128, 577
    s = 2 if dtype == torch.bfloat16 else 4
    gen = torch.Generator(device=dev).manual_seed(7)
    seqs = [torch.randn(B, N, D, device=dev, generator=gen).to(dtype) for _ in range(6)]
    w = torch.rand(B, N, device=dev, generator=gen)
    out = {}
    for keep in (0.25, 0.5, 0.75, 0.9):
        K = ops.keep_count(N, 1 - keep)
        for _ in range(3):
            idx, _b = ops.topk_indices(w, K, need_bottom=False)
            ops.gather_mix(seqs[0], idx, 0.05)
        reps = 24
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            idx, _b = ops.topk_indices(w, K, need_bottom=False)
            ops.gather_mix(seqs[r % len(seqs)], idx, 0.05)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = B * (N * D * s + N * 4 + (K + 1) * D * s + K * 8)
        out[f"keep_{keep}"] = {"ms": round(ms, 5), "GB/s": round(nbytes / ms / 1e6, 1),
                               "frac_of_8TBs": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)}
    return out


def cpu_baseline(sample):
    """The oracle (reference op sequence: MHA weights -> topk -> gather -> masked mean -> cat ->
    mean; per-sample gradients by vmap(grad) -> norms -> clip -> sum -> noise -> scale) on the
    host cores, fp32, one bounded batch."""
    from oracle import dp_oracle as do
    from oracle import fusion_oracle as fo
    from oracle import inputs as gi
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    B = sample
    v_seq, v_global = gi.tokens(1, B, N_VIS)
    q_seq, q_global = gi.tokens(2, B, N_TXT)
    t2v, v2t = gi.mha_params(3), gi.mha_params(7)
    y = gi.labels(5, B, C)
    cls = do.make_classifier(2 * D, 1024, C).eval()
    k = fo.keep_count(N_VIS, 1 - KEEP)
    kq = fo.keep_count(N_TXT, 1 - KEEP)
    vd = gi.adversarial_draws(11, B, k, N_VIS - k, fo.adv_count(k, 0.2))
    qd = gi.adversarial_draws(12, B, kq, N_TXT - kq, fo.adv_count(kq, 0.2))

    def one():
        vs, qs = v_seq.clone().requires_grad_(), q_seq.clone().requires_grad_()
        fused, _, _ = fo.co_attention_pruned(vs, v_global, qs, q_global,
                                             (t2v["in_proj_weight"], t2v["in_proj_bias"]),
                                             (v2t["in_proj_weight"], v2t["in_proj_bias"]),
                                             v_draws=vd, q_draws=qd)
        f = fused.detach().requires_grad_()
        nn.functional.cross_entropy(cls(f), y).backward()
        fused.backward(f.grad)  # input gradients of the fusion
        do.dp_step(cls, fused.detach(), y, 1.0, 1.0, LOGICAL_BATCH)

    one()
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 10.0:
        one()
        reps += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(B / dt, 2), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{reps} steps of batch {B} (same shapes; fp32; oracle = reference op sequence, "
                      f"per-sample grads by vmap)"}


def run_job(args, dev, dtype, rank, world, local_b, logical_b, probe):
    """Builds the job at the given per-rank batch, captures it, times `steps` steps.
    Returns (seconds for the timed steps (max over ranks), launch mode, eager ms/step or None)."""
    from vqa_amd import dist as vdist
    fus, gsm, opt, main_opt, sets = build_job(dev, local_b, dtype, rank, logical_b)
    step = make_step(fus, gsm, opt, main_opt, sets)
    # eager warm-up: creates every lazily built buffer (flat gradient bucket, RNG states, RCCL
    # communicators) before anything is captured
    for i in range(max(args.warmup, len(sets))):
        step(i)
    torch.cuda.synchronize()
    graphs = g_apply = None
    if not args.eager and (world == 1 or args.graph_multi):
        # the launch-bound step (about 60 short kernels) is captured once per input set into a HIP
        # graph; replays draw fresh noise because the RNG counters live in device memory
        # With several ranks the collective stays OUT of the graphs: one graph per input set for
        # the work before it, one for the update after it, and a plain RCCL all-reduce between.
        try:
            graphs, pool = [], None
            for i in range(len(sets)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    if world == 1:
                        step(i)
                    else:
                        step.local(i)
                pool = g.pool()
                graphs.append(g)
            g_apply = None
            if world > 1:
                g_apply = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_apply, pool=pool):
                    step.apply()
            for g in graphs:
                g.replay()
                if g_apply is not None:
                    step.reduce()
                    g_apply.replay()
            torch.cuda.synchronize()
        except Exception as exc:  # capture is an optimisation of the launch path only
            if rank == 0:
                print(f"[bench] HIP graph capture unavailable ({type(exc).__name__}: {exc}); "
                      "timing eager launches", file=sys.stderr)
            graphs = g_apply = None
            torch.cuda.synchronize()

    def run(i):
        if graphs is None:
            step(i)
        else:
            graphs[i % len(graphs)].replay()
            if g_apply is not None:
                step.reduce()
                g_apply.replay()

    for i in range(args.warmup):
        run(i)
    vdist.barrier()
    torch.cuda.synchronize()
    if probe is not None:
        probe.enabled = graphs is None
    t0 = time.perf_counter()
    for i in range(args.steps):
        run(args.warmup + i)
    torch.cuda.synchronize()
    vdist.barrier()
    dt = time.perf_counter() - t0
    eager_ms = None
    if probe is not None:
        probe.enabled = False
        if graphs is not None:
            # stream events cannot be read back from inside a replayed graph, so the prune+mix
            # kernel is timed over the same number of eagerly launched steps right afterwards
            probe.enabled = True
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(args.warmup + i)
            torch.cuda.synchronize()
            eager_ms = (time.perf_counter() - t1) / args.steps * 1e3
            probe.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, ("hipgraph" if graphs is not None else "eager"), eager_ms


def main():
    args = parse()
    if os.environ.get("VQAF_BENCH_WATCHDOG"):  # developer aid: dump all stacks if a rehearsal stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VQAF_BENCH_WATCHDOG"]), exit=True)
    from vqa_amd import dist as vdist
    from vqa_amd import ops
    rank, local, world = vdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    dev = vdist.local_device(local)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.scaling == "weak":
        local_b, logical_b = args.batch, args.batch * world
    else:
        lo, hi = vdist.shard_range(args.batch, rank, world)
        local_b, logical_b = hi - lo, args.batch

    probe = Probe("gather_mix_pool_multi")
    ops.PROBE = probe
    dt, launch, eager_ms = run_job(args, dev, dtype, rank, world, local_b, logical_b, probe)
    ops.PROBE = None
    other = None
    if world > 1:  # the other reading of configs[2]: same logical batch 1024, sharded
        if args.scaling == "weak":
            lo, hi = vdist.shard_range(args.batch, rank, world)
            o_local, o_logical, o_name = hi - lo, args.batch, "strong_scaling_1024"
        else:
            o_local, o_logical, o_name = args.batch, args.batch * world, "weak_scaling_1024_per_gpu"
        o_dt, o_launch, _ = run_job(args, dev, dtype, rank, world, o_local, o_logical, None)
        other = (o_name, {"value": round(o_logical * args.steps / o_dt, 1), "unit": "samples/s",
                          "global_batch": o_logical, "per_gpu_batch": o_local,
                          "ms_per_step": round(o_dt / args.steps * 1e3, 4), "launch": o_launch})

    if rank == 0:
        s = 2 if dtype == torch.bfloat16 else 4
        K = ops.keep_count(N_VIS, 1 - KEEP)
        # algorithmic bytes of the pooled prune+mix launch (vision and text tokens in one launch):
        # every token row read once, the keep indices read, one pooled row written per modality
        Kt = ops.keep_count(N_TXT, 1 - KEEP)
        alg_bytes = local_b * ((N_VIS * D * s + K * 8 + D * s) + (N_TXT * D * s + Kt * 8 + D * s))
        k_ms = probe.mean_ms()
        achieved = alg_bytes / k_ms / 1e6 if k_ms > 0 else 0.0
        traffic = None
        try:  # HBM bytes per launch from the PMC passes kept under profiles/ (same shape only)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            if local_b == 1024 and args.dtype == "bf16":
                traffic = pmc["kernels"]["gather_mix_pool_multi_kernel<true, true, 32>"]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
        line = {
            "metric": "DP-SGD train samples/s (fusion+classifier+DP step, token-level inputs)",
            "value": round(logical_b * args.steps / dt, 1),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "launch": launch,
            "ms_per_step_eager": round(eager_ms, 4) if eager_ms is not None else None,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: DP-SGD step, C=1.0, sigma=1.0, "
                                   f"logical batch {logical_b} = {local_b} per GPU x {world} GPU(s); "
                                   f"CoAttention pruning keep 0.5, {N_VIS} vision + {N_TXT} text tokens x {D}, "
                                   f"{C} classes; encoders bypassed",
                       "global_batch": logical_b, "per_gpu_batch": local_b, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "kernel": "gather_mix_pool_multi_kernel (prune+mix, vision + text tokens)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(k_ms, 5), "algorithmic_bytes": alg_bytes},
        }
        if other is not None:
            line[other[0]] = other[1]
        if not args.no_sweep:
            line["prune_mix_cfg4"] = prune_mix_sweep(dev, dtype)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
