"""Encoder-shaped Linear GEMMs: the library's own bf16 kernel against torch's (developer tool).

    python tools/encoder_gemm_probe.py [tokens]

Prints, per shape, the time and TFLOP/s of forward (y = x W^T + b) and input gradient (dx = dy W) on both.
"""
import sys

import torch
import torch.nn.functional as F

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 197
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (out_f, in_f) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        x = torch.randn(T, in_f, device=dev).bfloat16()
        w = (torch.randn(out_f, in_f, device=dev) * 0.02).bfloat16()
        b = torch.randn(out_f, device=dev).bfloat16()
        gy = torch.randn(T, out_f, device=dev).bfloat16()
        y = torch.empty(T, out_f, dtype=torch.bfloat16, device=dev)
        gx = torch.empty(T, in_f, dtype=torch.bfloat16, device=dev)
        flop = 2.0 * T * out_f * in_f
        rows = []
        if ops.gemm_bf16_supported(in_f, False, in_f, False, T, out_f, in_f):
            t = timed(lambda: ops.gemm_bf16(x, False, w, False, T, out_f, in_f, y, bias=b))
            rows.append(("own fwd", t))
            ref = F.linear(x, w, b)
            err = (y.float() - ref.float()).abs().max().item()
        else:
            err = float("nan")
        rows.append(("lib fwd", timed(lambda: F.linear(x, w, b))))
        if ops.linear_nt_supported(x, w):
            y2 = ops.linear_nt(x, w, b)
            e2 = (y2.float() - F.linear(x, w, b).float()).abs().max().item()
            rows.append((f"big fwd (err {e2:.3g})", timed(lambda: ops.linear_nt(x, w, b))))
            aux = torch.empty(T, out_f, dtype=torch.bfloat16, device=dev)
            rows.append(("big fwd+gelu", timed(lambda: ops.linear_nt(x, w, b, False, 1, aux))))
            gx2 = ops.linear_nt(gy, w, None, True)
            e4 = (gx2.float() - torch.mm(gy, w).float()).abs().max().item()
            rows.append((f"big dgrad (err {e4:.3g})", timed(lambda: ops.linear_nt(gy, w, None, True))))
        if ops.gemm_bf16_supported(out_f, False, in_f, True, T, in_f, out_f):
            rows.append(("own dgrad", timed(lambda: ops.gemm_bf16(gy, False, w, True, T, in_f, out_f, gx))))
        rows.append(("lib dgrad", timed(lambda: torch.mm(gy, w))))
        if ops.gemm_bf16_supported(out_f, True, in_f, True, out_f, in_f, T):
            direct = ((out_f + 127) // 128) * ((in_f + 63) // 64) >= 192
            gw = torch.empty(out_f, in_f, dtype=torch.bfloat16 if direct else torch.float32, device=dev)
            cs = torch.empty(out_f, dtype=torch.float32, device=dev)
            rows.append(("own wgrad", timed(lambda: ops.gemm_bf16(gy, True, x, True, out_f, in_f, T, gw, colsum=cs))))
        rows.append(("lib wgrad", timed(lambda: torch.mm(gy.t(), x))))
        if ops.wgrad_tn_supported(gy, x):
            c, cs2 = ops.wgrad_tn(gy, x, torch.float32, torch.float32)
            ref = gy.double().t() @ x.double()
            e5 = ((c.double() - ref).abs().max() / ref.abs().max()).item()
            e6 = ((cs2.double() - gy.double().sum(0)).abs().max() / gy.double().sum(0).abs().max()).item()
            rows.append((f"big wgrad (rel err {e5:.2g}, bias {e6:.2g})", timed(lambda: ops.wgrad_tn(gy, x))))
        print(f"T={T} out={out_f} in={in_f} (fwd max err {err:.3g}):\n   " +
              "\n   ".join(f"{n} {t * 1e3:.0f} us {flop / t / 1e9:.0f} TF" for n, t in rows), flush=True)


if __name__ == "__main__":
    main()
