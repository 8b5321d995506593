"""Training / validation loops with the call order of the reference (utils/src.py:24-126, 129-245)
and no per-batch host synchronisation.

The reference calls `.item()` three times per training step (utils/src.py:109-111) and copies
predictions to the host every validation batch; here loss and accuracy are accumulated on the
device and copied once per epoch.  Signatures follow the reference (`train(model, train_loader,
optimizer, criterion, device, model_type, use_dp_sgd, lambda_ib)`; `optimizer` is a torch
optimizer or the dict {'main': ..., 'classifier': DPOptimizer} that train.py:162 builds).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import dist as vdist
from . import ops


def _batch(batch, device):
    inputs = batch['inputs']
    return (batch['image'].to(device), inputs['input_ids'].to(device), inputs['attention_mask'].to(device),
            batch['answer'].to(device))


def _main_reducer(optimizer) -> "vdist.GradReducer | None":
    """The gradient exchange of a NON-private optimizer in a process group (cached on the optimizer);
    None with one rank, and for a DP optimizer, which reduces its own clipped sums."""
    if hasattr(optimizer, "reduce_step"):
        return None
    red = getattr(optimizer, "_vqaf_reducer", None)
    if red is None:
        red = vdist.GradReducer([p for g in optimizer.param_groups for p in g["params"]])
        optimizer._vqaf_reducer = red
    return red if red.world() > 1 else None


def train(model: nn.Module, train_loader, optimizer, criterion, device, model_type: str = "VQAModel",
          use_dp_sgd: bool = False, lambda_ib: float = 0.01) -> Tuple[float, float]:
    """One epoch; returns (average loss, accuracy in %) like utils/src.py:123-126.

    In a process group (one rank per GPU, rank-disjoint batches) every replica ends a step with the
    same parameters: the non-private optimizer steps on the batch-weighted mean of all ranks'
    gradients (vqa_amd.dist.GradReducer), the DP optimizer on the all-reduced clipped sums."""
    if model_type == "VQAModel_IB":
        raise NotImplementedError("VQAModel_IB is outside the accelerated path")
    model.train()
    split = isinstance(optimizer, dict)
    main_opt = optimizer['main'] if split else optimizer
    reducer = _main_reducer(main_opt)
    tot_loss = torch.zeros((), dtype=torch.float64, device=device)
    tot_correct = torch.zeros((), dtype=torch.int64, device=device)
    n = 0
    for batch in train_loader:
        if batch is None:  # an empty Poisson-sampled batch
            # the other ranks are waiting in the collectives, in this order: non-private gradients,
            # then the classifier's clipped sums.  This rank contributes zeros to the first and takes
            # the same averaged step as everyone else; Opacus still takes (and accounts for) the
            # noise-only step of the private part.
            if reducer is not None:
                main_opt.zero_grad()
                reducer.reduce(0)
                main_opt.step()
            if split and hasattr(optimizer['classifier'], 'empty_step'):
                optimizer['classifier'].empty_step(device)
            elif hasattr(optimizer, 'empty_step'):
                optimizer.empty_step(device)
            continue
        images, ids, mask, answers = _batch(batch, device)
        if split:                                   # utils/src.py:66-70
            optimizer['main'].zero_grad()
            optimizer['classifier'].zero_grad()
        else:
            optimizer.zero_grad()
        outputs = model(images=images, input_ids=ids, attention_mask=mask)   # :91-95
        loss = criterion(outputs, answers)                                   # :96
        loss.backward(gradient=ops.unit_gradient(loss))                      # :99 (the same unit gradient, cached)
        b = images.size(0)
        if reducer is not None:   # replicas: the mean over the ranks' joint batch, as one process would see it
            reducer.reduce(b)
        if split:                                                            # :102-106
            optimizer['main'].step()
            optimizer['classifier'].step()
        else:
            optimizer.step()
        tot_loss += loss.detach().double() * b                              # :109 without .item()
        tot_correct += (outputs.detach().argmax(dim=1) == answers).sum()    # :110
        n += b
    if n == 0:
        return 0.0, 0.0
    return float(tot_loss.item() / n), float(tot_correct.item() / n * 100.0)


@torch.no_grad()
def validate(model: nn.Module, val_loader, criterion, device, model_type: str = "VQAModel",
             mode: str = "val") -> Tuple[float, float, Dict[str, torch.Tensor]]:
    """Returns (average loss, accuracy in %, {'preds': ..., 'labels': ...}); the reference's
    sklearn / seaborn reporting (utils/src.py:183-243) is left to the caller."""
    model.eval()
    tot_loss = torch.zeros((), dtype=torch.float64, device=device)
    preds, labels = [], []
    n = 0
    for batch in val_loader:
        if batch is None:
            continue
        images, ids, mask, answers = _batch(batch, device)
        out = model(images=images, input_ids=ids, attention_mask=mask)
        out = out[0] if isinstance(out, tuple) else out
        tot_loss += criterion(out, answers).double() * images.size(0)
        preds.append(out.argmax(dim=1))
        labels.append(answers)
        n += images.size(0)
    if n == 0:
        return 0.0, 0.0, {"preds": torch.empty(0), "labels": torch.empty(0)}
    p, l = torch.cat(preds), torch.cat(labels)
    acc = float((p == l).sum().item() / n * 100.0)
    return float(tot_loss.item() / n), acc, {"preds": p.cpu(), "labels": l.cpu()}
