"""Developer tool: library speed of the weight-gradient GEMM under the four operand layouts."""
import os, sys
import torch
dev = torch.device("cuda:0")
T = 12608
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for out_f, in_f in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    gy = torch.randn(T, out_f, device=dev).to(torch.bfloat16)
    x = torch.randn(T, in_f, device=dev).to(torch.bfloat16)
    gyT = gy.t().contiguous(); xT = x.t().contiguous()
    fl = 2.0 * T * out_f * in_f
    r = {}
    r["kmaj,kmaj"] = timeit(lambda: torch.mm(gy.t(), x))
    r["kmaj,kcon"] = timeit(lambda: torch.mm(gy.t(), xT.t()))
    r["kcon,kmaj"] = timeit(lambda: torch.mm(gyT, x))
    r["kcon,kcon"] = timeit(lambda: torch.mm(gyT, xT.t()))
    r["T(gy)"] = timeit(lambda: gy.t().contiguous())
    r["T(x)"] = timeit(lambda: x.t().contiguous())
    r["dWt kmaj,kmaj"] = timeit(lambda: torch.mm(x.t(), gy))
    print(out_f, in_f, " ".join("%s=%.0fus(%.0fTF)" % (k, v, fl / v / 1e6) for k, v in r.items()))
