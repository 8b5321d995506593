import sys, torch
sys.path.insert(0, '/root/repo')
from vqa_amd._lib import check, load, ptr, stream_ptr
dev = torch.device('cuda:0'); lib = load(); st = stream_ptr(dev)
def timeit(fn, reps=50, warm=5):
    for i in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N in (30, 64, 197, 577):
    K = N // 2
    for B in (1, 64, 256, 1024, 4096):
        w = torch.rand(B, N, device=dev)
        top = torch.empty(B, K, dtype=torch.int64, device=dev); bot = torch.empty(B, N-K, dtype=torch.int64, device=dev)
        t = timeit(lambda: check(lib.vqaf_topk(ptr(w), B, N, K, ptr(top), ptr(bot), st), 'x'))
        print(f"topk N={N:4d} B={B:5d}: {t:7.2f} us", flush=True)
x = torch.zeros(2, dtype=torch.int64, device=dev)
t = timeit(lambda: check(lib.vqaf_rng_advance(ptr(x), 1, st), 'x'))
print(f"rng_advance (launch floor): {t:.2f} us")
