"""Where a workgroup of the one-pass fused kernel spends its time (developer tool).

Needs the stamped build: `make -C vqa_amd/csrc stamps` (tools/bin/libvqafuse_stamps.so).  In that build
thread 0 of every workgroup writes ten s_memrealtime stamps (100 MHz) into the sample's row of the
`w_out` buffer.  Prints, per modality, the median duration of every phase and of a workgroup's life,
and the span of the whole launch.
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqa_amd import _lib, ops  # noqa: E402

_lib._lib = _lib.load(os.path.join(ROOT, "tools", "bin", "libvqafuse_stamps.so"))
lib = _lib._lib
dev = torch.device("cuda:0")
B, D, H = int(os.environ.get("B", 1024)), 512, 8
NV, NQ = int(os.environ.get("NV", 197)), int(os.environ.get("NQ", 30))
NOISE = os.environ.get("NOISE")
gen = torch.Generator(device=dev).manual_seed(1)
nbuf = 4
vs = [torch.randn(B, NV, D, device=dev, generator=gen).bfloat16() for _ in range(nbuf)]
qs = [torch.randn(B, NQ, D, device=dev, generator=gen).bfloat16() for _ in range(nbuf)]
w = torch.randn(3 * D, D, device=dev) / D ** 0.5
bias = torch.randn(3 * D, device=dev) * 0.1
qg = torch.randn(B, D, device=dev).bfloat16()
rng = ops.PhiloxState(dev, seed=3)
st = _lib.stream_ptr(dev)
qp = torch.empty(B, D, device=dev)
u = torch.empty(B, H, D, device=dev)
_lib.check(lib.vqaf_fold_query(_lib.ptr(qg), _lib.ptr(w), _lib.ptr(bias), B, D, H, _lib.BF16, _lib.F32, _lib.ptr(qp),
                               _lib.ptr(u), st), "fold")
out = torch.empty(B, 2 * D, device=dev, dtype=torch.bfloat16)
NAMES = ["issue loads + stage u", "barrier", "scoring (waits for rows)", "barrier", "selection", "multiplicities + weights",
         "pooling MFMAs", "index store + noise", "final fold + store"]


def run(which):
    mods, keep = [], []
    for i, (seqs, N) in enumerate(((vs, NV), (qs, NQ))):
        if which not in ("both", ("vision", "text")[i]):
            continue
        k = ops.keep_count(N, 0.5)
        ka = ops.adv_count(k, 0.2)
        idx = torch.empty(B, k, dtype=torch.int64, device=dev)
        stamps = torch.zeros(B, 48, device=dev)  # 24 u64 per sample
        kind = _lib.NOISE_GAUSSIAN if (NOISE and i == 0) else _lib.NOISE_NONE
        keep.append((idx, stamps, N))
        mods.append(lambda it, seqs=seqs, idx=idx, stamps=stamps, N=N, k=k, ka=ka, kind=kind, i=i: _lib.FusedMod(
            _lib.ptr(seqs[it % nbuf]), _lib.ptr(u), None, None, None, _lib.ptr(rng.state), _lib.ptr(idx),
            _lib.ptr(stamps), i, 0.05, 0.01, kind, N, k, 2, ka, i * D, 1))
    n = len(mods)
    for it in range(6):
        arr = (_lib.FusedMod * n)(*[m(it) for m in mods])
        _lib.check(lib.vqaf_fused_prune_pool_onepass_multi(ctypes.cast(arr, ctypes.c_void_p), n, B, D, H, _lib.BF16,
                                                   _lib.ptr(rng.state), 0, _lib.ptr(out), 2 * D, _lib.BF16, st), "fused")
    torch.cuda.synchronize()
    t_all = []
    for idx, stamps, N in keep:
        full = stamps.cpu().numpy().reshape(-1).view(np.uint64).reshape(B, 24).astype(np.int64)
        t = full[:, :10]
        t_all.append(t)
        d = np.diff(t, axis=1) / 100.0  # us
        print(f"  N={N}: workgroup life median {np.median(t[:, 9] - t[:, 0]) / 100:.1f} us; phases (median us): "
              + ", ".join(f"{nm} {np.median(d[:, j]):.2f}" for j, nm in enumerate(NAMES)))
        sel = np.concatenate([full[:, 4:5], full[:, 10:16], full[:, 5:6]], axis=1)
        ds = np.diff(sel, axis=1) / 100.0
        print("     selection: " + ", ".join(f"{nm} {np.median(ds[:, j]):.2f}" for j, nm in enumerate(
            ["softmax stats", "weights", "keys", "ranking", "swap prep + position draw", "source draw", "swap"])))
        hw = full[:, 18] & 0xffffffff
        xcc = full[:, 18] >> 32
        keys = ((xcc & 15) << 8) | ((hw >> 8) & 0xff)
        print(f"     stagger: {int(full[:, 16].sum())} of {B} workgroups waited; distinct compute-unit keys {len(np.unique(keys))}; "
              f"xcc values {sorted(set(xcc.tolist()))[:10]}; start gap percentiles (10 ns) "
              f"{np.percentile(full[:, 17].astype(np.float64), [1, 10, 50, 90]).tolist()}")
    t = np.concatenate(t_all)
    print(f"  launch span (first start to last end) {(t[:, 9].max() - t[:, 0].min()) / 100:.1f} us", flush=True)


for which in ("vision", "text", "both"):
    print(which)
    run(which)
