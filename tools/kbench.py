"""Per-kernel timing of the libvqafuse entry points at the benchmark shapes (developer tool).

    python tools/kbench.py [--B 1024 --N 197] [--only name,...]

Each kernel is timed with stream events over `reps` launches on rotating input buffers (so the
Infinity Cache does not serve the tokens) and reported with its algorithmic bytes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops  # noqa: E402
from vqa_amd._lib import check, dtype_code, load, ptr, stream_ptr  # noqa: E402


ISOLATED = False


def timeit(fn, reps=30, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    if ISOLATED:  # one launch at a time on an idle GPU: no overlap between consecutive launches
        tot = 0.0
        for i in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(i)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1024)
    ap.add_argument("--N", type=int, default=197)
    ap.add_argument("--D", type=int, default=512)
    ap.add_argument("--keep", type=float, default=0.5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--isolated", action="store_true", help="synchronise between launches")
    a = ap.parse_args()
    global ISOLATED
    ISOLATED = a.isolated
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    s = 2 if a.dtype == "bf16" else 4
    B, N, D, H = a.B, a.N, a.D, 8
    K = ops.keep_count(N, 1 - a.keep)
    nbuf = max(2, int(600e6 // (B * N * D * s)) + 1)
    seqs = [torch.randn(B, N, D, device=dev).to(dt) for _ in range(nbuf)]
    w = torch.rand(B, N, device=dev)
    idx, bottom = ops.topk_indices(w, K)
    lib, st = load(), stream_ptr(dev)
    only = set(x for x in a.only.split(",") if x)
    res = []

    def run(name, nbytes, fn):
        if only and name not in only:
            return
        ms = timeit(fn)
        res.append((name, ms, nbytes / ms / 1e6))
        print(f"{name:28s} {ms*1e3:9.2f} us  {nbytes/ms/1e6:9.1f} GB/s  ({nbytes/1e6:.1f} MB)", flush=True)

    print(f"B={B} N={N} D={D} K={K} dtype={a.dtype} buffers={nbuf}")
    # ceilings on this box: device copy and a plain token mean with the same geometry
    dsts = [torch.empty_like(seqs[0]) for _ in range(2)]
    run("torch copy (r+w)", 2 * B * N * D * s, lambda i: dsts[i % 2].copy_(seqs[i % nbuf]))
    mp = torch.empty(B, D, dtype=dt, device=dev)
    run("mean_pool (read only)", B * N * D * s,
        lambda i: check(lib.vqaf_mean_pool(ptr(seqs[i % nbuf]), B, N, D, dtype_code(seqs[0]), ptr(mp), D, 0,
                                           dtype_code(mp), st), "x"))
    # pooled prune+mix forward / backward
    out = torch.empty(B, 2 * D, dtype=dt, device=dev)
    run("gather_mix_pool", B * (N * D * s + K * 8 + D * s),
        lambda i: check(lib.vqaf_gather_mix_pool(ptr(seqs[i % nbuf]), ptr(idx), None, 0, 0.0, None, 0.05, B, N, K, D,
                                                 dtype_code(seqs[0]), ptr(out), 2 * D, 0, dtype_code(out), st), "x"))
    # both modalities of CoAttention in one launch (what the step uses)
    import ctypes
    from vqa_amd import _lib as L
    Nt = 30
    Kt = ops.keep_count(Nt, 1 - a.keep)
    qseqs = [torch.randn(B, Nt, D, device=dev).to(dt) for _ in range(nbuf)]
    qidx, _qb = ops.topk_indices(torch.rand(B, Nt, device=dev), Kt)
    arrs = []
    for i in range(nbuf):
        arr = (L.PoolMod * 2)()
        arr[0] = L.PoolMod(ptr(seqs[i]), ptr(idx), None, None, 0.0, 0.05, 0, N, K, 0)
        arr[1] = L.PoolMod(ptr(qseqs[i]), ptr(qidx), None, None, 0.0, 0.05, 0, Nt, Kt, D)
        arrs.append(arr)
    run("gather_mix_pool_multi", B * ((N * D * s + K * 8 + D * s) + (Nt * D * s + Kt * 8 + D * s)),
        lambda i: check(lib.vqaf_gather_mix_pool_multi(ctypes.cast(arrs[i % nbuf], ctypes.c_void_p), 2, B, D,
                                                       dtype_code(seqs[0]), ptr(out), 2 * D, dtype_code(out), st), "x"))
    gouts = [torch.randn(B, 2 * D, device=dev).to(dt) for _ in range(2)]
    gseqs = [torch.empty(B, N, D, dtype=dt, device=dev) for _ in range(nbuf)]
    run("gather_mix_pool_bwd", B * (N * D * s + K * 8 + D * s),
        lambda i: check(lib.vqaf_gather_mix_pool_bwd(ptr(gouts[i % 2]), 2 * D, 0, dtype_code(gouts[0]), ptr(idx), 0.05,
                                                     B, N, K, D, dtype_code(gseqs[0]), ptr(gseqs[i % nbuf]), st), "x"))
    # unfused module op
    outs = [torch.empty(B, K + 1, D, dtype=dt, device=dev) for _ in range(nbuf)]
    run("gather_mix", B * (N * D * s + K * 8 + (K + 1) * D * s),
        lambda i: check(lib.vqaf_gather_mix(ptr(seqs[i % nbuf]), ptr(idx), None, 0, 0.0, None, 0.05, B, N, K, D,
                                            dtype_code(seqs[0]), dtype_code(outs[0]), ptr(outs[i % nbuf]), st), "x"))
    rng = ops.PhiloxState(dev, seed=1)
    run("gather_mix+gauss_noise", B * (N * D * s + K * 8 + (K + 1) * D * s),
        lambda i: check(lib.vqaf_gather_mix(ptr(seqs[i % nbuf]), ptr(idx), None, 1, 0.01, ptr(rng.state), 0.05, B, N, K,
                                            D, dtype_code(seqs[0]), dtype_code(outs[0]), ptr(outs[i % nbuf]), st), "x"))
    gos = [torch.randn(B, K + 1, D, device=dev).to(dt) for _ in range(nbuf)]
    run("gather_mix_bwd", B * ((K + 1) * D * s + N * D * s + K * 8),
        lambda i: check(lib.vqaf_gather_mix_bwd(ptr(gos[i % nbuf]), ptr(idx), 0.05, B, N, K, D, dtype_code(gos[0]),
                                                dtype_code(gseqs[0]), ptr(gseqs[i % nbuf]), st), "x"))
    # scoring
    u = torch.randn(B, H, D, device=dev)
    sc = torch.empty(B, N, H, device=dev)
    run("head_scores", B * (N * D * s + N * H * 4),
        lambda i: check(lib.vqaf_head_scores(ptr(seqs[i % nbuf]), ptr(u), B, N, D, H, dtype_code(seqs[0]), ptr(sc), st),
                        "x"))
    qu = torch.randn(B, H, D, device=dev)
    qsc = torch.empty(B, Nt, H, device=dev)
    sds = []
    for i in range(nbuf):
        sd = (L.ScoresDesc * 2)()
        sd[0] = L.ScoresDesc(ptr(seqs[i]), ptr(u), ptr(sc), N)
        sd[1] = L.ScoresDesc(ptr(qseqs[i]), ptr(qu), ptr(qsc), Nt)
        sds.append(sd)
    run("head_scores_multi", B * ((N + Nt) * D * s + (N + Nt) * H * 4),
        lambda i: check(lib.vqaf_head_scores_multi(ctypes.cast(sds[i % nbuf], ctypes.c_void_p), 2, B, D, H,
                                                   dtype_code(seqs[0]), st), "x"))
    wout = torch.empty(B, N, device=dev)
    run("attn_weights", B * (N * H * 4 + N * 4),
        lambda i: check(lib.vqaf_attn_weights(ptr(sc), B, N, H, ptr(wout), st), "x"))
    top = torch.empty(B, K, dtype=torch.int64, device=dev)
    bot = torch.empty(B, N - K, dtype=torch.int64, device=dev)
    run("topk", B * (N * 4 + N * 8),
        lambda i: check(lib.vqaf_topk(ptr(w), B, N, K, ptr(top), ptr(bot), st), "x"))
    k_adv = ops.adv_count(K, 0.2)
    if N - K >= k_adv:
        run("adv_swap_random", B * k_adv * 16,
            lambda i: check(lib.vqaf_adv_swap_random(ptr(top), ptr(bot), B, K, N - K, k_adv, ptr(rng.state), None, None, st), "x"))
    q = torch.randn(B, D, device=dev).to(dt)
    wq = torch.randn(3 * D, D, device=dev).to(dt)
    bq = torch.randn(3 * D, device=dev).to(dt)
    qp = torch.empty(B, D, device=dev)
    run("fold_query", 2 * D * D * s + B * D * s + B * H * D * 4,
        lambda i: check(lib.vqaf_fold_query(ptr(q), ptr(wq), ptr(bq), B, D, H, dtype_code(q), dtype_code(wq), ptr(qp), ptr(u),
                                            st), "x"))
    # DP step kernels at the classifier's shapes
    for (In, Out) in ((1024, 2048), (2048, 1024), (1024, 430)):
        aa = torch.randn(B, In, device=dev).to(dt)
        gg = torch.randn(B, Out, device=dev).to(dt)
        coef = torch.rand(B, device=dev)
        SW = torch.empty(Out, In, device=dev)
        Sb = torch.empty(Out, device=dev)
        run(f"clipped_sum {Out}x{In}", B * (In + Out) * s + Out * In * 4,
            lambda i: check(lib.vqaf_clipped_sum_linear(ptr(aa), ptr(gg), ptr(coef), B, In, Out, dtype_code(aa), 0, ptr(SW),
                                                        ptr(Sb), st), "x"))
    return res


if __name__ == "__main__":
    main()
