"""Developer tool: library GEMM times for the encoder shapes with PyTorch's TunableOp search on."""
import os, sys, time
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = "/tmp/tunableop.csv"
os.environ["PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS"] = "30"
os.environ["PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS"] = "5"
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
    w = torch.randn(out_f, in_f, device=dev).to(torch.bfloat16)
    fl = 2.0 * T * out_f * in_f
    t0 = time.time()
    r = {"wgrad": timeit(lambda: torch.mm(gy.t(), x)), "fwd": timeit(lambda: torch.nn.functional.linear(x, w)),
         "dgrad": timeit(lambda: torch.mm(gy, w))}
    print(out_f, in_f, " ".join("%s=%.0fus(%.0fTF)" % (k, v, fl / v / 1e6) for k, v in r.items()), "tuning %.1fs" % (time.time() - t0), flush=True)
