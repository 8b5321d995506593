"""CPU restatement of the DP-SGD step the reference delegates to Opacus -- TEST INFRASTRUCTURE.

PARITY UNPINNED by the reference: the arithmetic lives in the third-party package
opacus==1.4.0 (requirements.txt:39 of the reference), which is neither vendored under
/root/reference nor installed, and the reference holds no test or golden vector for it.  What is
restated here is Opacus 1.4.0's published algorithm at the reference's call sites
(train.py:124-153; utils/src.py:99,104):

  * ModuleValidator.fix: BatchNorm1d(n) -> GroupNorm(min(32, n), n, affine)       (train.py:124)
  * GradSampleModule, loss_reduction="mean": per-sample gradients g_i of the UN-averaged loss
    (backprops are multiplied by the batch size)
  * DPOptimizer.clip_and_accumulate: n_i = ||g_i||_2 over all parameters,
    c_i = min(1, C / (n_i + 1e-6)), S = sum_i c_i g_i
  * DPOptimizer.add_noise: S + N(0, (sigma C)^2)
  * DPOptimizer.scale_grad: grad = (S + noise) / expected_batch_size

Per-sample gradients are computed literally (torch.func.vmap(grad)) -- the slow, obviously
correct way -- so that the ghost-norm / clipped-sum kernels are checked against an independent
formulation.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.func import functional_call, grad, vmap


def make_classifier(in_dim: int = 1024, hidden_dim: int = 1024, num_classes: int = 430,
                    dp_fixed: bool = True) -> nn.Sequential:
    """The classifier of model/model.py:54-63; with dp_fixed the two BatchNorm1d layers are the
    GroupNorm layers ModuleValidator.fix would put there."""
    def norm(n):
        return nn.GroupNorm(min(32, n), n, affine=True) if dp_fixed else nn.BatchNorm1d(n)
    return nn.Sequential(
        nn.Linear(in_dim, hidden_dim * 2), norm(hidden_dim * 2), nn.ReLU(), nn.Dropout(0.3),
        nn.Linear(hidden_dim * 2, hidden_dim), norm(hidden_dim), nn.ReLU(),
        nn.Linear(hidden_dim, num_classes))


def per_sample_grads(model: nn.Module, x: torch.Tensor, y: torch.Tensor) -> Dict[str, torch.Tensor]:
    """g_i = grad of CE(model(x_i), y_i) for every i; dict name -> (B, *param.shape).
    The model must be in eval mode or have no active dropout (vmap cannot replay torch's
    dropout stream); tests keep dropout off or feed a fixed mask through `model`."""
    params = {k: v.detach() for k, v in model.named_parameters()}
    buffers = {k: v.detach() for k, v in model.named_buffers()}

    def loss_one(p, xi, yi):
        out = functional_call(model, (p, buffers), (xi.unsqueeze(0),))
        return F.cross_entropy(out, yi.unsqueeze(0))

    return vmap(grad(loss_one), in_dims=(None, 0, 0))(params, x, y)


def dp_step(model: nn.Module, x: torch.Tensor, y: torch.Tensor, max_grad_norm: float,
            noise_multiplier: float, expected_batch_size: int,
            noise: Optional[Dict[str, torch.Tensor]] = None
            ) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """Returns (norms (B,), clip factors (B,), clipped sums S, final grads).
    `noise[name]` is a standard-normal tensor per parameter (multiplied here by sigma*C)."""
    gs = per_sample_grads(model, x, y)
    B = x.shape[0]
    sq = torch.zeros(B, dtype=torch.float64)
    for g in gs.values():
        sq += g.reshape(B, -1).double().pow(2).sum(dim=1)
    norms = sq.sqrt().float()
    clip = (max_grad_norm / (norms + 1e-6)).clamp(max=1.0)
    sums, grads = {}, {}
    for name, g in gs.items():
        s = torch.einsum("i,i...->...", clip, g)
        sums[name] = s
        z = noise[name] if noise is not None else torch.zeros_like(s)
        grads[name] = (s + noise_multiplier * max_grad_norm * z) / expected_batch_size
    return norms, clip, sums, grads
