"""Seeded synthetic inputs shared by the golden-vector generator, the tests, smoke() and bench.py.

Everything is drawn with numpy's PCG64 generator (a stream that does not depend on the torch
build or the device), so a fixture only has to store the seed and the expected outputs.
TEST INFRASTRUCTURE: not imported by vqa_amd/.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch


def normal(seed: int, *shape: int, scale: float = 1.0) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    # .clone(): torch-allocated (64-byte aligned) storage; CPU GEMM kernels pick their code path
    # by operand alignment, and numpy's 16-byte-aligned buffers change result bits.
    return torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * np.float32(scale))).clone()


def tokens(seed: int, B: int, N: int, D: int = 512) -> Tuple[torch.Tensor, torch.Tensor]:
    """(seq (B,N,D), global (B,D) = seq.mean(1)) -- the token-level recipe of the benchmark."""
    seq = normal(seed, B, N, D)
    return seq, seq.mean(dim=1)


def mha_params(seed: int, E: int = 512) -> Dict[str, torch.Tensor]:
    """in_proj / out_proj tensors with the names nn.MultiheadAttention uses."""
    s = 1.0 / np.sqrt(E)
    return {
        "in_proj_weight": normal(seed, 3 * E, E, scale=s),
        "in_proj_bias": normal(seed + 1, 3 * E, scale=0.1),
        "out_proj.weight": normal(seed + 2, E, E, scale=s),
        "out_proj.bias": normal(seed + 3, E, scale=0.1),
    }


def adversarial_draws(seed: int, B: int, k: int, remaining_k: int, k_adv: int
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per sample: the first k_adv entries of a permutation of [0,k) and of [0,remaining_k)."""
    rng = np.random.default_rng(seed)
    pos = np.stack([rng.permutation(k)[:k_adv] for _ in range(B)])
    src = np.stack([rng.permutation(remaining_k)[:k_adv] for _ in range(B)])
    return torch.from_numpy(pos.astype(np.int64)), torch.from_numpy(src.astype(np.int64))


def classifier_params(seed: int, in_dim: int, hidden_dim: int, num_classes: int) -> Dict[str, torch.Tensor]:
    """State dict of the (GroupNorm-fixed) classifier, names as in model/model.py:54-63."""
    h2 = hidden_dim * 2
    return {
        "0.weight": normal(seed, h2, in_dim, scale=1.0 / np.sqrt(in_dim)),
        "0.bias": normal(seed + 1, h2, scale=0.05),
        "1.weight": 1.0 + normal(seed + 2, h2, scale=0.1),
        "1.bias": normal(seed + 3, h2, scale=0.1),
        "4.weight": normal(seed + 4, hidden_dim, h2, scale=1.0 / np.sqrt(h2)),
        "4.bias": normal(seed + 5, hidden_dim, scale=0.05),
        "5.weight": 1.0 + normal(seed + 6, hidden_dim, scale=0.1),
        "5.bias": normal(seed + 7, hidden_dim, scale=0.1),
        "7.weight": normal(seed + 8, num_classes, hidden_dim, scale=1.0 / np.sqrt(hidden_dim)),
        "7.bias": normal(seed + 9, num_classes, scale=0.05),
    }


def labels(seed: int, B: int, num_classes: int) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.integers(0, num_classes, size=B).astype(np.int64))
