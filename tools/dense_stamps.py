"""Where a workgroup of token_bwd_multi_kernel spends its time (developer tool).  Needs the stamped build:
`make -C vqa_amd/csrc stamps`.  Thread 0 of every workgroup writes four s_memrealtime stamps (100 MHz):
start, coefficients ready, token gradient written, folded-query gradient done.  args: B N"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqa_amd import _lib  # noqa: E402

_lib._lib = _lib.load(os.path.join(ROOT, "tools", "bin", "libvqafuse_stamps.so"))
lib = _lib._lib
B, N = int(sys.argv[1]), int(sys.argv[2])
D, H, NQ = 512, 8, 30
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
f32 = dict(dtype=torch.float32, device=dev)
Ns = (N, NQ)
seqs = [torch.randn(B, n, D, device=dev, generator=gen).bfloat16() for n in Ns]
ps = [torch.softmax(torch.randn(B, H, n, device=dev, generator=gen), -1) for n in Ns]
dps = [torch.randn(B, n, H, device=dev, generator=gen) for n in Ns]
us = [torch.randn(B, H, D, device=dev, generator=gen) for _ in Ns]
ts = [torch.randn(B, H, D, device=dev, generator=gen) for _ in Ns]
du = [torch.empty(B, H, D, **f32) for _ in Ns]
gs = [torch.empty_like(s) for s in seqs]
bm = (_lib.DenseBwdMod * 2)(*[_lib.DenseBwdMod(_lib.ptr(seqs[i]), _lib.ptr(ps[i]), _lib.ptr(dps[i]), _lib.ptr(us[i]),
                                               _lib.ptr(ts[i]), _lib.ptr(du[i]), _lib.ptr(gs[i]), Ns[i]) for i in range(2)])
st = _lib.stream_ptr(dev)
FWD = len(sys.argv) > 3 and sys.argv[3] == "fwd"  # stamps of softmax_wsum_multi_kernel instead (same slots:
# start, coefficients ready, (unused), weighted sum done; 4/5 = scores staged / maximum reduced)
if FWD:
    ss = [torch.randn(B, n, H, device=dev, generator=gen) for n in Ns]
    zs = [torch.empty(B, H, D, **f32) for _ in Ns]
    fm = (_lib.DenseFwdMod * 2)(*[_lib.DenseFwdMod(_lib.ptr(seqs[i]), _lib.ptr(ss[i]), _lib.ptr(ps[i]), _lib.ptr(zs[i]), Ns[i])
                                  for i in range(2)])
for _ in range(3):
    if FWD:
        _lib.check(lib.vqaf_softmax_wsum_multi(ctypes.cast(fm, ctypes.c_void_p), 2, B, D, H, _lib.BF16, None, 0, st), "fwd")
    else:
        _lib.check(lib.vqaf_token_bwd_multi(ctypes.cast(bm, ctypes.c_void_p), 2, B, D, H, _lib.BF16, st), "bwd")
torch.cuda.synchronize()
nwg = 4 * B * 2
buf = (ctypes.c_ulonglong * (nwg * 8))()
lib.vqaf_debug_xa_stamps.restype = ctypes.c_int
lib.vqaf_debug_xa_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.vqaf_debug_xa_stamps(buf, min(nwg, 8192) * 8) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)[:min(nwg, 8192)].astype(np.float64) / 100.0  # us
a = raw[:, :4]
print("vision prologue: loads + LDS stores %.1f us, block reduction %.1f us, ds %.1f us (medians)" % (
    np.median(raw[:4 * B, 4] - raw[:4 * B, 0]), np.median(raw[:4 * B, 5] - raw[:4 * B, 4]),
    np.median(raw[:4 * B, 1] - raw[:4 * B, 5])))
t0 = a[:, 0].min()
for name, rows in (("vision", a[:4 * B]), ("text", a[4 * B:])):
    if len(rows) == 0:
        continue
    d = np.diff(rows, axis=1)
    print(f"{name}: workgroups {len(rows)}  start (min/median/max) {rows[:, 0].min() - t0:.1f} / "
          f"{np.median(rows[:, 0]) - t0:.1f} / {rows[:, 0].max() - t0:.1f} us;  "
          f"coefficients {np.median(d[:, 0]):.1f}  token gradient {np.median(d[:, 1]):.1f}  "
          f"folded-query gradient {np.median(d[:, 2]):.1f} us;  end max {rows[:, 3].max() - t0:.1f}")
