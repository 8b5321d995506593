"""CPU: host-side logic of the drop-in modules (no kernels run)."""
import pytest
import torch

from vqa_amd import fusion, ops


def test_keep_and_adv_counts():
    assert ops.keep_count(577, 0.5) == 288 and ops.keep_count(197, 0.5) == 98
    assert ops.keep_count(577, 0.1) == 519 and ops.keep_count(4, 0.99) == 1
    assert ops.adv_count(288, 0.2) == 57 and ops.adv_count(2, 0.2) == 1


def test_module_surface_matches_reference_names():
    m = fusion.CoAttention(use_pruning=True, prune_ratio=0.25, noise_injection=False)
    assert m.output_dim == 1024 and m.use_pruning
    assert m.v_pruning.prune_ratio == 0.25 and not m.q_pruning.noise_injection
    assert m.q_pruning.noise_scale == 0
    keys = sorted(m.state_dict().keys())
    assert keys == sorted(
        f"{a}.{k}" for a in ("txt_to_vis_attn", "vis_to_txt_attn")
        for k in ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias"))
    a = fusion.Attention(use_pruning=True)
    assert a.pruning.noise_injection and a.pruning.noise_scale == 0.01  # class defaults leak in
    assert sorted(a.state_dict()) == sorted(
        f"attention.{k}" for k in ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias"))
    c = fusion.Concat(512, 512)
    assert c.output_dim == 1024
    x = torch.ones(2, 512)
    assert c(None, x, None, 2 * x).shape == (2, 1024)


def test_kernels_refuse_cpu_tensors():
    """There is no CPU fallback: CPU inputs fail loudly instead of taking another path."""
    m = fusion.CoAttention(use_pruning=True)
    v = torch.randn(2, 49, 512)
    q = torch.randn(2, 30, 512)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(v, v.mean(1), q, q.mean(1))


def test_unknown_noise_type_raises_like_reference():
    p = fusion.PrivacyAwareTokenPruning(noise_type="laplace")
    with pytest.raises(ValueError, match="Unsupported noise type"):
        p._noise_args(None)
