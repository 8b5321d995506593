"""Developer tool: weight-gradient GEMMs of the encoder layers, own bf16 kernel vs the BLAS library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import ops
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 12608

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for out_f, in_f in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (512, 768)]:
    dy = torch.randn(T, out_f, device=dev).to(torch.bfloat16)
    x = torch.randn(T, in_f, device=dev).to(torch.bfloat16)
    dw32 = torch.empty(out_f, in_f, device=dev)
    dw16 = torch.empty(out_f, in_f, device=dev, dtype=torch.bfloat16)
    cs = torch.empty(out_f, device=dev)
    t_lib = timeit(lambda: torch.mm(dy.t(), x))
    t_lib_b = timeit(lambda: dy.sum(0))
    t32 = timeit(lambda: ops.gemm_bf16(dy, True, x, True, out_f, in_f, T, dw32, colsum=cs))
    t16 = timeit(lambda: ops.gemm_bf16(dy, True, x, True, out_f, in_f, T, dw16))
    ref = (dy.float().t() @ x.float())
    err = float((dw32 - ref).norm() / ref.norm())
    fl = 2.0 * T * out_f * in_f
    print("wgrad %4dx%4d K=%d: lib %.1f us (%.0f TF) + bias sum %.1f | own f32 out %.1f us (%.0f TF) bf16 out %.1f us  err %.1e"
          % (out_f, in_f, T, t_lib, fl / t_lib / 1e6, t_lib_b, t32, fl / t32 / 1e6, t16, err))
    w = torch.randn(out_f, in_f, device=dev).to(torch.bfloat16)
    t_f = timeit(lambda: torch.nn.functional.linear(x, w))
    t_d = timeit(lambda: torch.mm(dy, w))
    print("   fwd lib %.1f us (%.0f TF)  dgrad lib %.1f us (%.0f TF)" % (t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6))
