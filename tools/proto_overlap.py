"""Developer experiment (timing only, results are racy): weight-gradient GEMMs deferred to a second stream, joined
once at the end of the backward pass."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import encoders, ops, model as vm

dev = torch.device("cuda:0")
AUX = torch.cuda.Stream(device=dev)
state = {"queued": False}

def _join():
    torch.cuda.current_stream(dev).wait_stream(AUX)
    state["queued"] = False

def backward(ctx, gy):
    x, w = ctx.saved_tensors
    out_f, in_f = w.shape
    gy2 = gy.reshape(-1, out_f).contiguous()
    x2 = x.reshape(-1, in_f).contiguous()
    T = x2.shape[0]
    cur = torch.cuda.current_stream(dev)
    AUX.wait_stream(cur)
    with torch.cuda.stream(AUX):
        direct = ((out_f + 127) // 128) * ((in_f + 63) // 64) >= 192
        gw = torch.empty(out_f, in_f, dtype=torch.bfloat16 if direct else torch.float32, device=x.device)
        cs = torch.empty(out_f, dtype=torch.float32, device=x.device)
        ops.gemm_bf16(gy2, True, x2, True, out_f, in_f, T, gw, colsum=cs)
        if not direct:
            gw = gw.to(w.dtype)
        gb = cs.to(ctx.bdtype)
    gy2.record_stream(AUX); x2.record_stream(AUX)
    if not state["queued"]:
        state["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_join)
    gx = torch.mm(gy2, w).view(x.shape) if ctx.needs_input_grad[0] else None
    return gx, gw, gb

if len(sys.argv) > 1 and sys.argv[1] == "on":
    encoders._LinearFn.backward = staticmethod(backward)
B = 64
torch.manual_seed(0)
m = vm.VQAModel(encoders.ViT, encoders.Bert, fusion_type="co_attention", num_classes=430, use_token_pruning=True,
                prune_ratio=0.5).to(dev).to(torch.bfloat16).train()
if len(sys.argv) > 2 and sys.argv[2] == "nobert":
    m.overlap_encoders = False
imgs = torch.randn(B, 3, 224, 224, device=dev, dtype=torch.bfloat16)
ids = torch.randint(1, 30000, (B, 30), device=dev); mask = torch.ones(B, 30, dtype=torch.long, device=dev)
y = torch.randint(0, 430, (B,), device=dev)
params = list(m.parameters())
def fb():
    for p in params: p.grad = None
    loss = ops.cross_entropy_mean(m(imgs, ids, mask), y)
    loss.backward(gradient=ops.unit_gradient(loss))
for _ in range(3): fb()
torch.cuda.synchronize()
def timeit(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("eager %.2f ms" % timeit(fb), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fb()
g.replay(); torch.cuda.synchronize()
print("graph %.2f ms" % timeit(g.replay), flush=True)
