"""Pure-write bandwidth of this GPU (developer tool): torch fill, memset and copy over 238 MB."""
import torch

dev = torch.device("cuda:0")
n = 1024 * 227 * 512
bufs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(4)]
src = torch.randn(n, device=dev).bfloat16()


def timed(fn, reps=20):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn, nbytes in (("fill_(1.0)", lambda i: bufs[i % 4].fill_(1.0), 2 * n), ("zero_()", lambda i: bufs[i % 4].zero_(), 2 * n),
                         ("copy_", lambda i: bufs[i % 4].copy_(src), 4 * n)):
    us = timed(fn)
    print(f"{name:12s} {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
