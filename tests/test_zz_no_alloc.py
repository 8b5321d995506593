"""vqaf_score_noalloc (fixed device scratch inside the library, no allocation) against vqaf_score
(caller workspace) through the C ABI: bit-identical weights on two calls in a row, the second
with a smaller input, so anything the first call left in the scratch array would show."""
import pytest
import torch

from oracle import inputs as gi

pytestmark = pytest.mark.gpu


def _both(lib, dev, B, N, dtype, seed):
    from vqa_amd._lib import F32, check, dtype_code, ptr, stream_ptr
    D, H = 512, 8
    seq, glob = gi.tokens(seed, B, N)
    p = gi.mha_params(seed + 1)
    seq_d, q_d = seq.to(dev).to(dtype), glob.to(dev).to(dtype)
    w_d, b_d = p["in_proj_weight"].to(dev), p["in_proj_bias"].to(dev)
    st = stream_ptr(dev)
    nbytes = lib.vqaf_score_workspace_bytes(B, N, D, H)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    want = torch.empty(B, N, device=dev)
    check(lib.vqaf_score(ptr(q_d), ptr(seq_d), ptr(w_d), ptr(b_d), B, N, D, H, dtype_code(seq_d), F32,
                         ptr(ws), nbytes, ptr(want), st), "vqaf_score")
    got = torch.full((B, N), float("nan"), device=dev)
    check(lib.vqaf_score_noalloc(ptr(q_d), ptr(seq_d), ptr(w_d), ptr(b_d), B, N, D, H, dtype_code(seq_d), F32,
                                 ptr(got), st), "vqaf_score_noalloc")
    torch.cuda.synchronize()
    return got.cpu(), want.cpu()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_score_noalloc_matches_workspace_variant(dtype):
    assert torch.cuda.is_available()
    from vqa_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    got, want = _both(lib, dev, 64, 577, dtype, 5)       # 64 x 577 tokens
    assert torch.equal(got, want)
    assert abs(float(got.sum(1).mean()) - 1.0) < 1e-5    # rows are head-averaged softmax weights
    got, want = _both(lib, dev, 3, 30, dtype, 9)         # a much smaller call right after
    assert torch.equal(got, want)
    # beyond the fixed scratch: refused on the host, nothing launched, output untouched
    from vqa_amd._lib import F32, ptr, stream_ptr
    big = lib.vqaf_score_workspace_bytes(4096, 577, 512, 8)
    assert big > (32 << 20)
    sentinel = torch.full((8,), 7.0, device=dev)
    rc = lib.vqaf_score_noalloc(ptr(sentinel), ptr(sentinel), ptr(sentinel), ptr(sentinel), 4096, 577, 512, 8,
                                _lib.BF16, F32, ptr(sentinel), stream_ptr(dev))
    assert rc == -6 and "scratch" in lib.vqaf_status_string(rc).decode()
    torch.cuda.synchronize()
    assert torch.equal(sentinel.cpu(), torch.full((8,), 7.0))
