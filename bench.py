"""Benchmark of the DP-SGD-trained fusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one DP-SGD training step of the fusion + classifier path on token-level synthetic
COCO-QA-shaped inputs already resident in HBM (encoders are outside the hot path):
zero_grad -> CoAttention(pruning, keep 0.5, adversarial mixing) forward -> classifier forward ->
mean CE -> backward (classifier input gradients, fusion input gradients) -> per-sample norms,
clip (C = 1.0), clipped sums, Gaussian noise (sigma = 1.0), RCCL all-reduce, AdamW step.

The workload is BASELINE.json configs[2]: logical batch 1024, sharded 1024/N per GPU (`--scaling
strong`, the default, because that is what the config names; at N = 1 the whole batch sits on one
GPU).  An N > 1 run also measures the weak-scaling variant (1024 per GPU, logical batch 1024*N: per-GPU
work fixed, the only exchange is the all-reduce of the classifier's noisy gradient sum) and reports it
under "weak_scaling_1024_per_gpu"; `--scaling weak` swaps the two.  At 128 samples per GPU the step is
bound by its launch count, so the strong-scaling figure flattens by construction.

Launch path: the step is captured into HIP graphs (one per rotating input set).  With several
ranks the collective stays outside the graphs: graph(local work) -> eager RCCL all-reduce ->
graph(update).  `--eager` launches every kernel from Python instead.

Rank 0 prints ONE JSON line: whole-job samples/s plus
  roofline      the pooled prune+mix kernel: algorithmic bytes / mean kernel time, against the
                8 TB/s spec peak and against a device-copy rate measured in the same run.  Kernel
                time comes from stream events on the launch stream around that one launch.  ROCm
                refuses event records inside a captured graph ("External events are disallowed"),
                so when the timed region replays graphs the events bracket the same kernel in the
                same number of eagerly launched steps run right AFTER the timed region
                ("kernel_timing" says which of the two applies);
  mfma          the bf16 MFMA clipped-sum kernels timed the same way: TFLOP/s and fraction of 2.5 PF;
  cpu_baseline  the oracle (the reference's op sequence) on the host cores, bounded sample;
  prune_mix_cfg4  the module-boundary prune+mix op of configs[3] (B = 128, N = 577, keep sweep).
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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak

LOGICAL_BATCH = 1024
N_VIS, N_TXT, D, C = 197, 30, 512, 430
KEEP = 0.5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=LOGICAL_BATCH,
                    help="logical whole-job batch (strong scaling) or per-GPU batch (weak scaling)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python (no HIP graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the fp32 tier sub-measurement")
    ap.add_argument("--no-full-model", action="store_true", help="skip the encoder + fusion forward/backward leg")
    ap.add_argument("--cpu-sample", type=int, default=64)
    return ap.parse_args()


class Probe:
    """Stream-event pairs around named kernel launches (torch's current stream is the stream the
    C ABI launches on).  `names` = the launches to time; a name ending in ':' matches by prefix."""

    def __init__(self, *names):
        self.names, self.pairs, self.enabled, self._open = names, {}, False, {}

    def _match(self, name):
        return any(name == n or (n.endswith(":") and name.startswith(n)) for n in self.names)

    def __call__(self, name, phase):
        if not self.enabled or not self._match(name):
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if phase == "begin":
            self._open[name] = ev
        else:
            self.pairs.setdefault(name, []).append((self._open.pop(name), ev))

    def mean_ms(self, name):
        pairs = self.pairs.get(name, [])
        return sum(a.elapsed_time(b) for a, b in pairs) / max(1, len(pairs))


def copy_rate_gbs(dev):
    """Device-to-device copy rate of this GPU in this run (read + written bytes / time): the
    practical ceiling a streaming kernel can be held against."""
    n = 256 << 20
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = [torch.empty_like(src) for _ in range(2)]
    for d in dst:
        d.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for r in range(reps):
        dst[r & 1].copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * reps / e0.elapsed_time(e1) / 1e6


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


def make_step(fus, gsm, opt, main_opt, sets, eager):
    """The training step in the order of utils/src.py:66-106 as vqa_amd.loops.GraphedStep runs it for
    any caller of the drop-in modules: zero_grad, forward, mean CE, backward, optimizer['main'] step,
    per-rank DP work, (RCCL all-reduces outside the graphs when there are several ranks), DP update.
    The rotating input sets stay where they are (`resident_inputs`): one graph per set, nothing copied."""
    from vqa_amd import loops, ops
    gstep = loops.GraphedStep(lambda v, vg, q, qg: gsm(fus(v, vg, q, qg)), {'main': main_opt, 'classifier': opt},
                              ops.cross_entropy_mean, resident_inputs=True, warmup=(1 << 30) if eager else 2)

    def step(i, force_eager=False):
        v, vg, q, qg, y = sets[i % len(sets)]
        if force_eager:
            return gstep.eager_step((v, vg, q, qg), y)
        return gstep((v, vg, q, qg), y)

    step.graphed = gstep
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

    def two_launch(seq, K):
        idx, _b = ops.topk_indices(w, K, need_bottom=False)
        ops.gather_mix(seq, idx, 0.05)

    def one_launch(seq, K):
        ops.prune_mix(seq, w, K, 0.05)

    def timed(fn, K):
        for _ in range(3):
            fn(seqs[0], K)
        reps = 24
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            fn(seqs[r % len(seqs)], K)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    fused_ok = ops.prune_mix_supported(N, D, dtype)
    for keep in (0.25, 0.5, 0.75, 0.9):
        K = ops.keep_count(N, 1 - keep)
        nbytes = B * (N * D * s + N * 4 + (K + 1) * D * s + K * 8)
        ms2 = timed(two_launch, K)
        ms = timed(one_launch, K) if fused_ok else ms2
        out[f"keep_{keep}"] = {"ms": round(ms, 5), "GB/s": round(nbytes / ms / 1e6, 1),
                               "frac_of_8TBs": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                               "launches": 1 if fused_ok else 2,
                               "two_launch_ms": round(ms2, 5)}
    return out


def one_pass_leg(dev):
    """The one-launch pruning branch at the step's shape (B = 1024, 197 + 30 tokens, bf16) on its two
    kernels, timed alone over rotating inputs: the two-pass kernel the step uses and the one-pass kernel
    (rows read once, held in registers / LDS, pooled on the matrix cores).  Bytes: one-pass algorithmic
    bytes N*D*s + H*D*4 + K*8 + D*s per sample and modality; fabric traffic from profiles/r03_pmc_fused.json."""
    from vqa_amd import ops
    B, H = LOGICAL_BATCH, 8
    gen = torch.Generator(device=dev).manual_seed(1)
    vs = [torch.randn(B, N_VIS, D, device=dev, generator=gen).bfloat16() for _ in range(4)]
    qs = [torch.randn(B, N_TXT, D, device=dev, generator=gen).bfloat16() for _ in range(4)]
    w = torch.randn(3 * D, D, device=dev, generator=gen) / D ** 0.5
    bias = torch.randn(3 * D, device=dev, generator=gen) * 0.1
    qg = torch.randn(B, D, device=dev, generator=gen).bfloat16()
    rng = ops.PhiloxState(dev, seed=3)

    def specs(i):
        out = []
        for seq, n in ((vs[i % 4], N_VIS), (qs[i % 4], N_TXT)):
            k = ops.keep_count(n, 1 - KEEP)
            out.append(dict(q=qg, seq=seq, in_proj_weight=w, in_proj_bias=bias, num_heads=H, k=k,
                            k_adv=ops.adv_count(k, 0.2), alpha=0.05))
        return out

    class P:
        def __init__(self):
            self.t, self.o = [], None

        def __call__(self, name, phase):
            if name != "fused_prune_pool":
                return
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            if phase == "begin":
                self.o = e
            else:
                self.t.append((self.o, e))

    K, Kt = ops.keep_count(N_VIS, 1 - KEEP), ops.keep_count(N_TXT, 1 - KEEP)
    per = lambda n, k: n * D * 2 + H * D * 4 + k * 8 + D * 2  # noqa: E731
    one_pass_bytes = B * (per(N_VIS, K) + per(N_TXT, Kt))
    try:
        with open(os.path.join(ROOT, "profiles", "r03_pmc_fused.json")) as fh:
            pmc = json.load(fh)["kernels"]
    except (OSError, ValueError, KeyError):
        pmc = {}
    out, saved_probe, saved_mode = {}, ops.PROBE, ops.FUSED_ONE_PASS
    try:
        for label, one, key in (("two_pass", False, "fused_prune_pool_kernel (both)"),
                                ("one_pass", True, "fused_onepass_kernel (both)")):
            ops.FUSED_ONE_PASS = one
            p = P()
            ops.PROBE = p
            with torch.no_grad():
                for i in range(24):
                    ops.fused_prune_pool(specs(i), torch.bfloat16, rng)
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in p.t[4:])
            ms = ts[len(ts) // 2]
            traffic = pmc.get(key, {}).get("fabric_bytes_per_launch")
            out[label] = {"ms": round(ms, 5), "one_pass_GB/s": round(one_pass_bytes / ms / 1e6, 1),
                          "frac_of_8TBs_on_one_pass_bytes": round(one_pass_bytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                          "fabric_bytes_per_launch": traffic,
                          "fabric_over_one_pass_bytes": round(traffic / one_pass_bytes, 3) if traffic else None}
    finally:
        ops.PROBE, ops.FUSED_ONE_PASS = saved_probe, saved_mode
    out["one_pass_bytes"] = one_pass_bytes
    return out


def dense_leg(dev, dtype):
    """The unpruned branch (dense single-query cross-attention, utils/fusion.py:257-268), forward and
    backward with token gradients, at configs[1]'s batch (64 x 197 tokens) and configs[3]'s
    (128 x 577).  Token traffic as the kernels are launched: scores and weighted sum read the tokens
    in the forward pass, both again in the backward pass, and the token gradient is written:
    5 * N * D * s per sample and modality (3 * N * D * s if every pass read the tokens once per
    direction)."""
    from vqa_amd import fusion
    s = 2 if dtype == torch.bfloat16 else 4
    out = {}
    for B, N in ((64, 197), (128, 577)):
        torch.manual_seed(1)
        m = fusion.CoAttention(embed_dim=D, num_heads=8, use_pruning=False).to(dev).to(dtype)  # as the step's fusion module
        gen = torch.Generator(device=dev).manual_seed(3)
        v = torch.randn(B, N, D, device=dev, generator=gen).to(dtype).requires_grad_()
        q = torch.randn(B, N_TXT, D, device=dev, generator=gen).to(dtype).requires_grad_()
        vg = torch.randn(B, D, device=dev, generator=gen).to(dtype)
        qg = torch.randn(B, D, device=dev, generator=gen).to(dtype)
        cot = torch.randn(B, 2 * D, device=dev, generator=gen)

        def fb():
            v.grad = None
            q.grad = None
            m.zero_grad(set_to_none=True)
            o = m(v, vg, q, qg)
            o.backward(cot.to(o.dtype))

        for _ in range(3):
            fb()
        torch.cuda.synchronize()
        launch, graph = "eager", None
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fb()
            launch = "hipgraph"
        except Exception as exc:  # capture is an optimisation of the launch path only
            print(f"[bench] dense leg: graph capture unavailable ({type(exc).__name__}); eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
        run = graph.replay if graph is not None else fb
        for _ in range(3):
            run()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = 5 * B * (N + N_TXT) * D * s
        out[f"B{B}_N{N}"] = {"ms_fwd_bwd": round(ms, 5), "launch": launch,
                             "token_GB/s": round(nbytes / ms / 1e6, 1),
                             "frac_of_8TBs": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                             "token_bytes": nbytes, "samples/s": round(B / ms * 1e3, 1)}
        del graph
    return out


def mia_leg(dev, dtype):
    """BASELINE configs[4] at one GPU's share of an 8-GPU run: the membership-inference scoring pass
    (attack/metric_src.py:318-426: per-sample loss and confidence in eval mode) over 25 000 token-level samples
    (200 000 / 8) in batches of 1000 -- fusion forward (pruning keep 0.5), classifier forward, one fused
    loss + confidence launch per batch -- and that launch alone over a shard's 25 000 x 430 logits."""
    from vqa_amd import dp, fusion, mia, ops
    torch.manual_seed(42)
    fus = fusion.CoAttention(embed_dim=D, num_heads=8, use_pruning=True, prune_ratio=1 - KEEP, noise_injection=False).to(dev)
    cls = nn.Sequential(nn.Linear(2 * D, 2048), nn.BatchNorm1d(2048), nn.ReLU(), nn.Dropout(0.3),
                        nn.Linear(2048, 1024), nn.BatchNorm1d(1024), nn.ReLU(), nn.Linear(1024, C))
    cls = dp.ModuleValidator.fix(cls).to(dev)
    if dtype == torch.bfloat16:
        fus, cls = fus.to(dtype), cls.to(dtype)
    fus.eval()
    cls.eval()
    shard, bs = 25000, 1000
    gen = torch.Generator(device=dev).manual_seed(7)
    v = torch.randn(bs, N_VIS, D, device=dev, generator=gen).to(dtype)
    q = torch.randn(bs, N_TXT, D, device=dev, generator=gen).to(dtype)
    vg, qg = v.float().mean(1).to(dtype), q.float().mean(1).to(dtype)
    y = torch.randint(0, C, (bs,), device=dev, generator=gen)

    def batch():
        with torch.no_grad():
            logits = cls(fus(v, vg, q, qg))
            loss, conf, _ = mia.score_logits(logits, y)
        return loss, conf

    for _ in range(3):
        batch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(shard // bs):
        batch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    logits = torch.randn(shard, C, device=dev, generator=gen).to(dtype)
    yy = torch.randint(0, C, (shard,), device=dev, generator=gen)
    for _ in range(3):
        mia.score_logits(logits, yy)
    e0.record()
    for _ in range(10):
        mia.score_logits(logits, yy)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / 10
    nbytes = shard * (C * logits.element_size() + 8 + 4 + 4 + 8)
    return {"workload": "configs[4], one GPU's shard of 8: 25 000 samples, batches of 1000, eager launches",
            "ms": round(ms, 3), "samples/s": round(shard / ms * 1e3, 1),
            "ce_conf_kernel_ms_25000x430": round(k_ms, 5), "ce_conf_GB/s": round(nbytes / k_ms / 1e6, 1)}


def transformers_encoders_ms(dev, vqa_model, imgs, ids, mask, y):
    """The same model and step with the two encoder stacks taken from the `transformers` package (what
    model/vision_encoder.py:97 and model/text_encoder.py:21 construct), eager launches, same GPU: the
    reference's own encoder path beside ours.  None when the package is not importable."""
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    try:
        import transformers
    except Exception:
        return None
    from vqa_amd import ops

    class _Vision(nn.Module):
        def __init__(self, out_features=512):
            super().__init__()
            self.vit = transformers.ViTModel(transformers.ViTConfig(), add_pooling_layer=True)
            self.seq_projection = nn.Linear(768, out_features)
            self.global_projection = nn.Linear(768, out_features)
            self.output_dim = out_features

        def forward(self, x):
            o = self.vit(pixel_values=x)
            return self.seq_projection(o.last_hidden_state), self.global_projection(o.pooler_output)

    class _Text(nn.Module):
        def __init__(self, hidden_dim=512):
            super().__init__()
            self.bert = transformers.BertModel(transformers.BertConfig())
            self.seq_projection = nn.Linear(768, hidden_dim)
            self.global_projection = nn.Linear(768, hidden_dim)
            self.output_dim = hidden_dim

        def forward(self, i, m):
            o = self.bert(input_ids=i, attention_mask=m)
            return self.seq_projection(o.last_hidden_state), self.global_projection(o.pooler_output)

    try:
        torch.manual_seed(0)
        m = vqa_model.VQAModel(_Vision, _Text, fusion_type="co_attention", num_classes=C, use_token_pruning=True,
                               prune_ratio=0.5).to(dev).to(torch.bfloat16).train()
        m.overlap_encoders = False  # the reference runs them one after the other
        params = list(m.parameters())

        def fb():
            for p in params:
                p.grad = None
            loss = ops.cross_entropy_mean(m(imgs, ids, mask), y)
            loss.backward(gradient=ops.unit_gradient(loss))

        for _ in range(3):
            fb()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fb()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    except Exception as exc:  # a reference-side failure must not take the benchmark line with it
        print(f"[bench] transformers encoder leg skipped ({type(exc).__name__}: {exc})", file=sys.stderr)
        return None


def full_model_leg(dev):
    """BASELINE configs[1]: the whole model (ViT-B/16 at 224 px + BERT-base at 30 tokens, CoAttention with
    token pruning at keep ratio 0.5, the classifier), forward + backward, batch 64, bf16 parameters and
    activations, non-private, random-init weights, synthetic inputs.  The GEMM flops counted are those of
    the two encoder stacks (three GEMMs of 2*T*in*out per Linear over forward and backward, attention
    4*T*N*E forward and 2.5x that backward)."""
    from vqa_amd import encoders, ops
    from vqa_amd import model as vqa_model
    B, S, L, E, LAYERS, FF = 64, 224, N_TXT, 768, 12, 3072
    torch.manual_seed(0)
    m = vqa_model.VQAModel(encoders.ViT, encoders.Bert, fusion_type="co_attention", num_classes=C,
                           use_token_pruning=True, prune_ratio=0.5).to(dev).to(torch.bfloat16).train()
    gen = torch.Generator(device=dev).manual_seed(5)
    imgs = torch.randn(B, 3, S, S, device=dev, generator=gen).to(torch.bfloat16)
    ids = torch.randint(1, 30000, (B, L), device=dev, generator=gen)
    mask = torch.ones(B, L, dtype=torch.long, device=dev)
    y = torch.randint(0, C, (B,), device=dev, generator=gen)
    params = [p for p in m.parameters()]

    def fb():
        for p in params:
            p.grad = None
        loss = ops.cross_entropy_mean(m(imgs, ids, mask), y)
        loss.backward(gradient=ops.unit_gradient(loss))

    for _ in range(3):
        fb()
    torch.cuda.synchronize()
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fb()
    e1.record()
    torch.cuda.synchronize()
    ms_eager = e0.elapsed_time(e1) / reps
    launch, graph = "eager", None
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fb()
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
        launch = "hipgraph"
    except Exception as exc:  # capture is an optimisation of the launch path only
        print(f"[bench] full-model leg: graph capture unavailable ({type(exc).__name__}: {exc}); eager", file=sys.stderr)
        graph = None
        torch.cuda.synchronize()
    ms, ms_graph = ms_eager, None
    if graph is not None:
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms_graph = e0.elapsed_time(e1) / reps
        # the headline is the faster of the two launch paths: eager launches overlap the two encoders on two
        # streams, a replayed graph saves the launch cost but runs its branches with less overlap
        if ms_graph <= ms_eager:
            ms = ms_graph
        else:
            launch = "eager"
    flops = 0.0
    for T, N in ((B * ((S // 16) ** 2 + 1), (S // 16) ** 2 + 1), (B * L, L)):
        gemm = 2.0 * T * (4 * E * E + 2 * E * FF) * LAYERS
        attn = 4.0 * T * N * E * LAYERS
        flops += 3.0 * gemm + 3.5 * attn
    del graph
    ref_ms = transformers_encoders_ms(dev, vqa_model, imgs, ids, mask, y)
    out_extra = {} if ref_ms is None else {
        "transformers_encoders_ms_fwd_bwd_eager": round(ref_ms, 3),
        "speedup_vs_transformers_encoders": round(ref_ms / ms_eager, 3)}
    return {**out_extra,
            "workload": "ViT-B/16 (224 px) + BERT-base (30 tokens) + CoAttention keep 0.5 + classifier, fwd+bwd, "
                        "batch 64, bf16, non-private, random init",
            "ms_fwd_bwd": round(ms, 3), "ms_fwd_bwd_eager": round(ms_eager, 3),
            "ms_fwd_bwd_hipgraph": None if ms_graph is None else round(ms_graph, 3), "launch": launch,
            "samples/s": round(B / ms * 1e3, 1), "encoder_TFLOP/s": round(flops / ms / 1e9, 1),
            "frac_of_2500_TFLOPs": round(flops / ms / 1e9 / 2500.0, 4)}


def cpu_baseline(sample):
    """The oracle (reference op sequence: MHA weights -> topk -> gather -> masked mean -> cat ->
    mean; per-sample gradients by vmap(grad) -> norms -> clip -> sum -> noise -> scale) on the
    host cores, fp32, one bounded batch."""
    from oracle import dp_oracle as do
    from oracle import fusion_oracle as fo
    from oracle import inputs as gi
    # the GPU box grants a one-GPU job 16 host CPUs and exports OMP_NUM_THREADS accordingly
    # (os.cpu_count() there reports the whole machine); elsewhere every visible core is used
    host_cpus = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", host_cpus))
    threads = max(1, min(threads, host_cpus))
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

    for _ in range(2):
        one()
    times, t_all = [], time.perf_counter()
    while len(times) < 5 or time.perf_counter() - t_all < 10.0:
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > 30.0:
            break
    times.sort()
    dt = times[len(times) // 2]
    return {"value": round(B / dt, 2), "unit": "samples/s", "cores": threads, "host_cpus": host_cpus,
            "kind": "port",
            "sample": f"median of {len(times)} steps of batch {B} after 2 warm-ups (same shapes; fp32; oracle = "
                      f"reference op sequence, per-sample grads by vmap)"}


def run_job(args, dev, dtype, rank, world, local_b, logical_b, probe):
    """Builds the job at the given per-rank batch, captures it, times `steps` steps.
    Returns (seconds for the timed steps (max over ranks), launch mode, eager ms/step or None)."""
    from vqa_amd import dist as vdist
    fus, gsm, opt, main_opt, sets = build_job(dev, local_b, dtype, rank, logical_b)
    step = make_step(fus, gsm, opt, main_opt, sets, args.eager)
    # Warm-up: the first two visits of an input set run eagerly (they create every lazily built buffer:
    # flat gradient bucket, RNG states, RCCL communicators), the third captures that set's HIP graph(s);
    # replays draw fresh noise because the RNG counters live in device memory.  With several ranks the
    # collectives stay OUT of the graphs (GraphedStep): graph(local work) -> eager all-reduce ->
    # graph(update), and the ranks agree on whether capture succeeded.
    for i in range(max(args.warmup, (4 if not args.eager else 1) * len(sets))):
        step(i)
    vdist.barrier()
    torch.cuda.synchronize()
    graphed = step.graphed.launch == "hipgraph"
    if probe is not None:
        probe.enabled = not graphed
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    vdist.barrier()
    dt = time.perf_counter() - t0
    eager_ms = None
    if probe is not None:
        probe.enabled = False
        if graphed:
            # event records are refused inside a captured graph on ROCm, so the probed kernels
            # are timed over the same number of eagerly launched steps right afterwards
            probe.enabled = True
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(args.warmup + i, force_eager=True)
            torch.cuda.synchronize()
            eager_ms = (time.perf_counter() - t1) / args.steps * 1e3
            probe.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, ("hipgraph" if graphed else "eager"), eager_ms


def roofline_obj(probe, local_b, dtype, launch, copy_gbs):
    """The `roofline` object of the dominant HBM-bound kernel of one measured job."""
    from vqa_amd import ops
    s = 2 if dtype == torch.bfloat16 else 4
    K = ops.keep_count(N_VIS, 1 - KEEP)
    Kt = ops.keep_count(N_TXT, 1 - KEEP)
    H = 8
    traffic = None
    fused = probe.mean_ms("fused_prune_pool") > 0
    if fused:
        # One launch scores, selects and pools both modalities.  Algorithmic bytes per sample and
        # modality in the two-pass accounting of a fused score+prune (top-k needs every score
        # before the first row can be pooled, so the tokens are read twice: N*D*s each pass),
        # plus the folded queries H*D*4, the keep indices K*8 and one pooled row D*s.
        # `one_pass_bytes` counts the tokens once (what HBM itself must deliver when the second
        # pass is served on-die).
        kname = "fused_prune_pool_kernel (scoring + selection + pooled prune/mix, vision + text tokens)"
        k_ms = probe.mean_ms("fused_prune_pool")
        per = lambda n, k, passes: passes * n * D * s + H * D * 4 + k * 8 + D * s  # noqa: E731
        alg_bytes = local_b * (per(N_VIS, K, 2) + per(N_TXT, Kt, 2))
        one_pass = local_b * (per(N_VIS, K, 1) + per(N_TXT, Kt, 1))
        pmc_file, pmc_key = "r03_pmc_fused.json", "fused_prune_pool_kernel (both)"
        pmc_field = "fabric_bytes_per_launch"
    else:
        # separate launches (f32 tokens, unsupported widths): the pooled prune+mix launch, every
        # token row read once, the keep indices read, one pooled row written per modality
        kname = "gather_mix_pool_multi_kernel (prune+mix, vision + text tokens)"
        k_ms = probe.mean_ms("gather_mix_pool_multi")
        alg_bytes = local_b * ((N_VIS * D * s + K * 8 + D * s) + (N_TXT * D * s + Kt * 8 + D * s))
        one_pass = alg_bytes
        pmc_file, pmc_key = "r01_pmc_traffic.json", "gather_mix_pool_multi_kernel<true, true, 32>"
        pmc_field = "hbm_bytes_per_launch"
    achieved = alg_bytes / k_ms / 1e6 if k_ms > 0 else 0.0
    # Fabric-side bytes per launch come from counter passes kept under profiles/ (rocprofv3 --pmc cannot
    # run inside this process); they describe ONE shape, so any other shape reports null and says why.
    # tools/pmc_fused.sh <tag> regenerates the file for the default shape.
    traffic_note = None
    try:
        with open(os.path.join(ROOT, "profiles", pmc_file)) as fh:
            pmc = json.load(fh)
        if local_b == 1024 and dtype == torch.bfloat16 and fused:
            traffic = pmc["kernels"][pmc_key][pmc_field]
            traffic_note = f"profiles/{pmc_file} (FETCH_SIZE x 2 + WRITE_SIZE, separate counter passes, same shape)"
        else:
            traffic_note = (f"null: profiles/{pmc_file} holds counters for batch 1024 bf16 only; this run is batch "
                            f"{local_b} {'bf16' if dtype == torch.bfloat16 else 'f32'}")
    except (OSError, KeyError, ValueError) as exc:
        traffic, traffic_note = None, f"null: {type(exc).__name__} reading profiles/{pmc_file}"
    timing = ("stream events around the launch in eagerly launched steps right after the timed region "
              "(the timed region replays HIP graphs)") if launch == "hipgraph" else \
             "stream events around the launch inside the timed region"
    return {"bound": "hbm", "kernel": kname,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note,
            "kernel_ms": round(k_ms, 5), "algorithmic_bytes": alg_bytes,
            "one_pass_bytes": one_pass,
            "one_pass_GBs": round(one_pass / k_ms / 1e6, 1) if k_ms > 0 else 0.0,
            "kernel_timing": timing, "copy_rate_same_run_GBs": round(copy_gbs, 1),
            "frac_of_copy_rate": round(achieved / copy_gbs, 4) if copy_gbs > 0 else None}


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

    probe = Probe("gather_mix_pool_multi", "fused_prune_pool", "clipped_sum_linear:")
    ops.PROBE = probe
    dt, launch, eager_ms = run_job(args, dev, dtype, rank, world, local_b, logical_b, probe)
    ops.PROBE = None
    other = None
    if world > 1:  # the other scaling mode over the same ranks
        if args.scaling == "weak":
            lo, hi = vdist.shard_range(args.batch, rank, world)
            o_local, o_logical, o_name = hi - lo, args.batch, "strong_scaling_1024"
        else:
            o_local, o_logical, o_name = args.batch, args.batch * world, "weak_scaling_1024_per_gpu"
        o_dt, o_launch, _ = run_job(args, dev, dtype, rank, world, o_local, o_logical, None)
        other = (o_name, {"value": round(o_logical * args.steps / o_dt, 1), "unit": "samples/s",
                          "global_batch": o_logical, "per_gpu_batch": o_local,
                          "ms_per_step": round(o_dt / args.steps * 1e3, 4), "launch": o_launch})

    f32_tier = None
    if world == 1 and dtype == torch.bfloat16 and not args.no_f32:
        # the reference's own precision (fp32 tokens, weights and tapes) over the same step, a
        # shorter run: the fusion goes through the separate f32 launches, the clipped sums through
        # the f32 matrix cores
        import copy
        a32 = copy.copy(args)
        a32.steps, a32.warmup = min(args.steps, 20), min(args.warmup, 5)
        p32 = Probe("gather_mix_pool_multi", "fused_prune_pool")
        ops.PROBE = p32
        dt32, launch32, _ = run_job(a32, dev, torch.float32, rank, world, local_b, logical_b, p32)
        ops.PROBE = None
        f32_tier = {"value": round(logical_b * a32.steps / dt32, 1), "unit": "samples/s",
                    "ms_per_step": round(dt32 / a32.steps * 1e3, 4), "steps": a32.steps, "launch": launch32,
                    "roofline": roofline_obj(p32, local_b, torch.float32, launch32, copy_rate_gbs(dev))}
    if rank == 0:
        copy_gbs = copy_rate_gbs(dev)
        mfma = []
        for name in sorted(probe.pairs):
            if not name.startswith("clipped_sum_linear:"):
                continue
            shapes = [tuple(int(v) for v in part.split("x")) for part in name.split(":")[1].split("+")]
            ms = probe.mean_ms(name)
            tf = sum(2.0 * local_b * o * i for o, i in shapes) / ms / 1e9 if ms > 0 else 0.0
            label = " + ".join(f"{o}x{i}" for o, i in shapes) + (" (one launch)" if len(shapes) > 1 else "")
            mfma.append({"kernel": f"clipped_sum_linear {label} over batch {local_b}", "ms": round(ms, 5),
                         "tflops": round(tf, 1), "frac_of_2.5PF": round(tf / MFMA_BF16_PEAK_TFLOPS, 4)})
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
            "roofline": roofline_obj(probe, local_b, dtype, launch, copy_gbs),
            "mfma": mfma,
        }
        if other is not None:
            line[other[0]] = other[1]
        if f32_tier is not None:
            line["f32"] = f32_tier
        if not args.no_sweep:
            line["prune_mix_cfg4"] = prune_mix_sweep(dev, dtype)
            line["dense_branch"] = dense_leg(dev, dtype)
            line["mia_cfg5"] = mia_leg(dev, dtype)
            if dtype == torch.bfloat16:
                line["one_pass_kernel"] = one_pass_leg(dev)
        if not args.no_full_model and dtype == torch.bfloat16 and world == 1:  # a one-GPU measurement
            line["full_model_cfg2"] = full_model_leg(dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
