import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import _lib, ops
from vqa_amd._lib import ptr, check, load, stream_ptr
dev = torch.device("cuda:0")
lib = load()
for B, C, K, groups, relu in ((256, 2048, 16, 32, 1), (100, 1024, 432, 32, 1), (64, 128, 16, 16, 0)):
    g = torch.Generator(device=dev).manual_seed(1)
    gy = torch.randn(B, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(K, C, device=dev, generator=g) / K ** 0.5).bfloat16()
    y = torch.relu(torch.randn(B, C, device=dev, generator=g)).bfloat16()
    xhat = torch.randn(B, C, device=dev, generator=g).bfloat16()
    rstd = torch.rand(B, groups, device=dev, generator=g) + 0.5
    ga = torch.randn(C, device=dev, generator=g).bfloat16()
    st = stream_ptr(dev)
    gxy = torch.empty(B, C, device=dev, dtype=torch.bfloat16)
    ops.gemm_bf16(gy, False, W, True, B, C, K, gxy)
    gp0 = torch.empty_like(y); gl0 = torch.empty_like(y)
    check(lib.vqaf_groupnorm_bwd(ptr(gxy), ptr(y), ptr(xhat), ptr(rstd), ptr(ga), B, C, groups, relu, 1.25 if relu else 1.0,
                                 _lib.BF16, ptr(gp0), ptr(gl0), st), "bwd")
    gp1 = torch.full_like(y, float("nan")); gl1 = torch.full_like(y, float("nan"))
    check(lib.vqaf_linear_dgrad_groupnorm_bwd(ptr(gy), K, ptr(W), C, B, C, K, ptr(y), ptr(xhat), ptr(rstd), ptr(ga), _lib.BF16,
                                              groups, relu, 1.25 if relu else 1.0, ptr(gp1), ptr(gl1), st), "fused")
    torch.cuda.synchronize()
    print(B, C, K, "g_pre equal", torch.equal(gp0, gp1), "gl equal", torch.equal(gl0, gl1),
          "max diff", float((gp0.float() - gp1.float()).abs().max()), float((gl0.float() - gl1.float()).abs().max()),
          "nan", int(torch.isnan(gl1.float()).sum()), flush=True)
    if not torch.equal(gl0, gl1):
        d = (gl0.float() - gl1.float()).abs()
        idx = torch.nonzero(d > 0)
        print("  first diffs", idx[:6].tolist(), "count", idx.shape[0])
