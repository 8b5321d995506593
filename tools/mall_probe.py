import sys, torch
sys.path.insert(0, '/root/repo')
from vqa_amd._lib import check, dtype_code, load, ptr, stream_ptr
dev = torch.device('cuda:0'); lib = load(); st = stream_ptr(dev)
def timeit(fn, reps=40, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, D = 197, 512
for B in (64, 128, 256, 512, 1024):
    x = torch.randn(B, N, D, device=dev).bfloat16()
    out = torch.empty(B, D, dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda i: check(lib.vqaf_mean_pool(ptr(x), B, N, D, 1, ptr(out), D, 0, 1, st), 'x'))
    nb = B * N * D * 2
    print(f"B={B:5d} {nb/1e6:7.1f} MB same buffer: {ms*1e3:7.2f} us {nb/ms/1e6:8.1f} GB/s")
