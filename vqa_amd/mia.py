"""Membership-inference scoring passes (attack/metric_src.py:318-365, 381-426 of the reference):
per-sample cross-entropy and max-softmax confidence over a loader, thresholded into member
predictions.  The reference copies every batch to the host; here scores stay on the device
(one fused kernel per batch for both statistics) and are copied back once at the end.  With
torch.distributed initialised, every rank scores its shard and the scores are all-gathered.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import ops


@torch.no_grad()
def score_logits(logits: torch.Tensor, targets: Optional[torch.Tensor] = None):
    """(loss (B,) or None, confidence (B,), prediction (B,)) from one fused kernel."""
    loss, conf, pred, _ = ops.ce_conf(logits, targets, want_loss=targets is not None,
                                      want_conf=True, want_pred=True)
    return loss, conf, pred


def _forward(model, batch, device):
    images = batch['image'].to(device)
    inputs = batch['inputs']
    out = model(images=images, input_ids=inputs['input_ids'].to(device),
                attention_mask=inputs['attention_mask'].to(device))
    return (out[0] if isinstance(out, tuple) else out), batch['answer'].to(device)


def _gather(x: torch.Tensor) -> torch.Tensor:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        sizes = [torch.zeros(1, dtype=torch.int64, device=x.device) for _ in range(dist.get_world_size())]
        dist.all_gather(sizes, torch.tensor([x.numel()], dtype=torch.int64, device=x.device))
        m = int(max(s.item() for s in sizes))
        pad = torch.zeros(m, dtype=x.dtype, device=x.device)
        pad[:x.numel()] = x
        parts = [torch.empty_like(pad) for _ in sizes]
        dist.all_gather(parts, pad)
        x = torch.cat([p[:int(s.item())] for p, s in zip(parts, sizes)])
    return x


@torch.no_grad()
def evaluate_privacy_loss(model, dataloader, device, threshold=1.0, is_member=True,
                          model_type="VQAModel") -> Dict[str, np.ndarray]:
    """attack/metric_src.py:318-365: loss <= threshold predicts "member"."""
    model.eval()
    chunks = []
    for batch in dataloader:
        logits, answers = _forward(model, batch, device)
        chunks.append(score_logits(logits, answers)[0])
    losses = _gather(torch.cat(chunks)).cpu().numpy()
    return {'losses': losses,
            'ground_truth': np.full(losses.shape, 1 if is_member else 0),
            'predictions': losses <= threshold}


@torch.no_grad()
def evaluate_privacy_confidence(model, dataloader, device, threshold=0.6, is_member=True,
                                model_type="VQAModel") -> Dict[str, np.ndarray]:
    """attack/metric_src.py:381-426: confidence >= threshold predicts "member"."""
    model.eval()
    chunks = []
    for batch in dataloader:
        logits, _ = _forward(model, batch, device)
        chunks.append(score_logits(logits, None)[1])
    conf = _gather(torch.cat(chunks)).cpu().numpy()
    return {'confidences': conf,
            'ground_truth': np.full(conf.shape, 1 if is_member else 0),
            'predictions': conf >= threshold}
