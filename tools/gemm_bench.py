"""Timing of vqaf_gemm_bf16 at the classifier's shapes against torch (hipBLASLt) (developer tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(layout, M, N, K):
    g = torch.Generator(device=dev).manual_seed(1)
    if layout == "fwd":
        A = torch.randn(M, K, device=dev, generator=g).bfloat16()
        B = torch.randn(N, K, device=dev, generator=g).bfloat16()
        ak, bk, ref = False, False, (lambda: torch.nn.functional.linear(A, B))
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    elif layout == "dgrad":
        A = torch.randn(M, K, device=dev, generator=g).bfloat16()
        B = torch.randn(K, N, device=dev, generator=g).bfloat16()
        ak, bk, ref = False, True, (lambda: A.matmul(B))
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    else:
        A = torch.randn(K, M, device=dev, generator=g).bfloat16()
        B = torch.randn(K, N, device=dev, generator=g).bfloat16()
        ak, bk, ref = True, True, (lambda: A.t().matmul(B))
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
    t = timeit(lambda: ops.gemm_bf16(A, ak, B, bk, M, N, K, out))
    tr = timeit(ref)
    fl = 2.0 * M * N * K
    print(f"{layout:6s} M={M:5d} N={N:5d} K={K:5d}  ours {t:7.1f} us ({fl/t/1e6:6.1f} TF)   torch {tr:7.1f} us ({fl/tr/1e6:6.1f} TF)",
          flush=True)


shapes = [("wgrad", 2048, 1024, 1024), ("wgrad", 1024, 2048, 1024), ("fwd", 1024, 2048, 1024), ("fwd", 1024, 1024, 2048),
          ("dgrad", 1024, 1024, 2048), ("dgrad", 1024, 2048, 1024)]
if len(sys.argv) > 1 and sys.argv[1] == "ksweep":
    shapes = [("wgrad", 2048, 1024, k) for k in (64, 256, 512, 1024, 2048, 4096)] + \
             [("fwd", 1024, 2048, k) for k in (64, 256, 1024, 4096)]
for s in shapes:
    case(*s)
