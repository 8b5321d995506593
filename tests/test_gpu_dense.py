"""GPU parity of the dense (non-pruning) cross-attention branch against the oracle and the
golden vectors captured from the reference's CoAttention(use_pruning=False), forward and backward
(inputs and every MHA parameter)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import fusion_oracle as fo
from oracle import inputs as gi

pytestmark = pytest.mark.gpu
KEYS = ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vqa_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _close(got, want, rtol=1e-4, atol_rel=1e-5):
    want = np.asarray(want)
    np.testing.assert_allclose(np.asarray(got), want, rtol=rtol, atol=atol_rel * max(1e-12, np.abs(want).max()))


@pytest.mark.parametrize("name", ["coattn_N49", "coattn_N197", "coattn_N577"])
def test_coattention_dense_vs_golden(dev, name):
    from vqa_amd import fusion
    g = load_golden(name)
    seed, B, N = int(g["seed"]), int(g["B"]), int(g["N"])
    v_seq, v_global = gi.tokens(seed, B, N)
    q_seq, q_global = gi.tokens(seed + 1, B, 30)
    cot = gi.normal(seed + 10, B, 1024)
    m = fusion.CoAttention(use_pruning=False)
    m.txt_to_vis_attn.load_state_dict(gi.mha_params(seed + 2))
    m.vis_to_txt_attn.load_state_dict(gi.mha_params(seed + 6))
    m = m.to(dev)
    vs, qs = v_seq.to(dev).requires_grad_(), q_seq.to(dev).requires_grad_()
    vg, qg = v_global.to(dev).requires_grad_(), q_global.to(dev).requires_grad_()
    out = m(vs, vg, qs, qg)
    _close(out.detach().cpu(), g["dense"])
    (out * cot.to(dev)).sum().backward()
    _close(vs.grad.sum(2).cpu(), g["dense_gv_rowsum"])
    _close(vs.grad.sum(1).cpu(), g["dense_gv_colsum"])
    _close(qs.grad.sum(2).cpu(), g["dense_gq_rowsum"])
    _close(qs.grad.sum(1).cpu(), g["dense_gq_colsum"])
    _close(vg.grad.cpu(), g["dense_gvglobal"])
    _close(qg.grad.cpu(), g["dense_gqglobal"])
    pg = {n: p.grad.cpu() for n, p in m.named_parameters()}
    _close(pg["txt_to_vis_attn.in_proj_weight"].sum(1), g["dense_g_t2v_in_w_rowsum"])
    _close(pg["txt_to_vis_attn.in_proj_bias"], g["dense_g_t2v_in_b"])
    _close(pg["txt_to_vis_attn.out_proj.weight"].sum(1), g["dense_g_t2v_out_w_rowsum"])
    _close(pg["txt_to_vis_attn.out_proj.bias"], g["dense_g_t2v_out_b"])
    _close(pg["vis_to_txt_attn.in_proj_weight"].sum(1), g["dense_g_v2t_in_w_rowsum"])
    _close(pg["vis_to_txt_attn.in_proj_bias"], g["dense_g_v2t_in_b"])


@pytest.mark.parametrize("E,H,N,B,dtype", [(512, 8, 197, 3, torch.float32), (256, 4, 50, 5, torch.float32),
                                           (64, 1, 9, 2, torch.float32), (512, 8, 300, 4, torch.bfloat16),
                                           (128, 2, 33, 70, torch.bfloat16)])
def test_attention_dense_full_gradients_vs_oracle(dev, E, H, N, B, dtype):
    """Attention(use_pruning=False): every gradient element against autograd on the oracle."""
    from vqa_amd import fusion
    v_seq, v_global = gi.tokens(40 + N, B, N, E)
    q_seq, q_global = gi.tokens(41 + N, B, 12, E)
    p = gi.mha_params(42 + N, E)
    cot = gi.normal(43 + N, B, 2 * E)
    if dtype == torch.bfloat16:
        v_seq, q_global = v_seq.bfloat16().float(), q_global.bfloat16().float()
    ref_p = {k: v.clone().requires_grad_() for k, v in p.items()}
    vr, qr = v_seq.clone().requires_grad_(), q_global.clone().requires_grad_()
    want = torch.cat([fo.single_query_attention(qr, vr, *(ref_p[k] for k in KEYS), H), qr], dim=1)
    (want * cot).sum().backward()

    m = fusion.Attention(embed_dim=E, num_heads=H, use_pruning=False)
    m.attention.load_state_dict(p)
    m = m.to(dev)
    vs = v_seq.to(dev).to(dtype).requires_grad_()
    qg = q_global.to(dev).to(dtype).requires_grad_()
    out = m(vs, v_global.to(dev).to(dtype), q_seq.to(dev).to(dtype), qg)
    (out.float() * cot.to(dev)).sum().backward()
    if dtype == torch.float32:
        rt, at = 1e-4, 2e-5
    else:
        rt, at = 1e-2, 1e-2
    _close(out.detach().float().cpu(), want.detach(), rt, at)
    _close(vs.grad.float().cpu(), vr.grad, rt, at)
    _close(qg.grad.float().cpu(), qr.grad, rt, at)
    for k in KEYS:
        _close(getattr_path(m.attention, k).grad.cpu(), ref_p[k].grad, rt, at)


def getattr_path(mod, path):
    for part in path.split("."):
        mod = getattr(mod, part)
    return mod
