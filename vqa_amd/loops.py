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
          use_dp_sgd: bool = False, lambda_ib: float = 0.01, capture: bool = False, pad_to: int = 0
          ) -> Tuple[float, float]:
    """One epoch; returns (average loss, accuracy in %) like utils/src.py:123-126.

    In a process group (one rank per GPU, rank-disjoint batches) every replica ends a step with the
    same parameters: the non-private optimizer steps on the batch-weighted mean of all ranks'
    gradients (vqa_amd.dist.GradReducer), the DP optimizer on the all-reduced clipped sums.

    capture=True replays the step as HIP graphs (GraphedStep, kept on the model between epochs): same
    results, without the per-kernel launch cost of Python; `pad_to` lets Poisson batches of varying
    size share graphs."""
    if model_type == "VQAModel_IB":
        raise NotImplementedError("VQAModel_IB is outside the accelerated path")
    model.train()
    split = isinstance(optimizer, dict)
    main_opt = optimizer['main'] if split else optimizer
    reducer = _main_reducer(main_opt)
    graphed = None
    if capture:
        graphed = getattr(model, "_vqaf_graphed", None)
        if graphed is None or graphed.optimizer is not optimizer:
            graphed = GraphedStep(lambda im, i, m: model(images=im, input_ids=i, attention_mask=m), optimizer, criterion,
                                  pad_to=pad_to)
            model._vqaf_graphed = graphed
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
        if graphed is not None:
            loss, outputs = graphed((images, ids, mask), answers)
            b = images.size(0)
            tot_loss += loss.detach().double() * b
            tot_correct += (outputs.detach().argmax(dim=1) == answers).sum()
            n += b
            continue
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


class _MaskedMeanCE(torch.autograd.Function):
    """sum_i m_i CE_i / B over a batch whose last rows are padding (m_i = 0): the padded rows get a
    zero logits gradient, so they contribute nothing to any parameter gradient or per-sample norm.
    `grad_fix` (0-d device tensor or None) multiplies the logits gradient: padded / real rows when no
    private module sits in between and every parameter wants the mean over the real rows."""

    @staticmethod
    def forward(ctx, logits, targets, mask, grad_fix):
        B = logits.shape[0]
        gdt = logits.dtype if logits.dtype in (torch.float32, torch.bfloat16) else torch.float32
        loss, _, _, grad = ops.ce_conf(logits, targets, want_loss=True, want_conf=False, grad_scale=1.0 / B, grad_dtype=gdt)
        w = mask if grad_fix is None else mask * grad_fix
        grad.mul_(w.to(grad.dtype).unsqueeze(1))
        ctx.save_for_backward(grad)
        ctx.in_dtype = logits.dtype
        return (loss * mask).sum() / B

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        if grad_out is ops._unit_grads.get((grad_out.device, grad_out.dtype)) and grad.dtype == ctx.in_dtype:
            return grad, None, None, None
        return (grad.float() * grad_out).to(ctx.in_dtype), None, None, None


class GraphedStep:
    """The training step of utils/src.py:66-106 -- zero_grad, forward, loss, backward, optimizer
    step(s) -- captured into HIP graphs and replayed, for batches of a static shape.

    forward(*inputs) -> logits; `optimizer` is a torch optimizer, a DPOptimizer, or the dict
    {'main': ..., 'classifier': DPOptimizer} of train.py:162; `criterion(logits, answers)` a mean loss.
    Launched kernel by kernel from Python the step is host-bound (tens of launches of a few
    microseconds each); replayed, its time is the sum of its kernels.  Random draws (adversarial
    mixing, token noise, dropout, DP noise) come from counters in device memory, so every replay
    draws fresh numbers and a replayed run equals the eagerly launched one.

    * One graph per input signature.  `resident_inputs=False`: the signature is the shapes; a batch is
      copied into the graph's own input buffers.  `resident_inputs=True`: the signature is the input
      tensors themselves (same storage every time, e.g. a few rotating device-resident sets) and
      nothing is copied.
    * The first `warmup` steps of a signature run eagerly (they are real steps and build every lazily
      created buffer); the next one is captured.  If capture fails the signature stays eager.
    * In a process group the collectives stay outside the graphs: graph(forward, backward, per-rank DP
      work) -> eager all-reduces (non-private gradients, clipped sums) -> graph(optimizer steps).
    * `pad_to`: batches of varying size (Poisson sampling) are padded with zero rows to a multiple of
      `pad_to` and a row mask keeps the padding out of the loss, the gradients and the per-sample
      norms, so a handful of graphs serve every batch size.  Needs a mean cross-entropy criterion and
      a model without batch statistics (the DP classifier's GroupNorm is per sample).
    """

    def __init__(self, forward, optimizer, criterion=None, *, resident_inputs: bool = False, warmup: int = 2,
                 pad_to: int = 0, max_graphs: int = 16):
        self.forward, self.optimizer = forward, optimizer
        self.criterion = criterion if criterion is not None else ops.cross_entropy_mean
        self.resident, self.warmup, self.pad_to, self.max_graphs = resident_inputs, int(warmup), int(pad_to), int(max_graphs)
        split = isinstance(optimizer, dict)
        self.dp_opt = optimizer['classifier'] if split else (optimizer if hasattr(optimizer, 'reduce_step') else None)
        self.main_opt = optimizer['main'] if split else (None if self.dp_opt is optimizer else optimizer)
        self.reducer = _main_reducer(self.main_opt) if self.main_opt is not None else None
        self.world = self.reducer.world() if self.reducer is not None else (
            self.dp_opt._world()[0] if self.dp_opt is not None else 1)
        ce_mean = self.criterion is ops.cross_entropy_mean or (
            isinstance(self.criterion, nn.CrossEntropyLoss) and self.criterion.reduction == "mean"
            and self.criterion.weight is None and self.criterion.label_smoothing == 0.0)
        if self.pad_to and not ce_mean:
            raise ValueError("pad_to needs a plain mean cross-entropy criterion")
        self._sigs: dict = {}
        self._pool = None
        self.launch = "eager"  # "hipgraph" once a signature replays

    # -- the step, piece by piece -----------------------------------------------------------------
    def _zero(self, inputs):
        for t in inputs:
            if t.requires_grad:
                t.grad = None
        if self.main_opt is not None:
            self.main_opt.zero_grad()
        if self.dp_opt is not None:
            self.dp_opt.zero_grad()

    def _loss(self, logits, answers, mask, fix):
        if mask is not None:
            # with a private module the factor is applied where the gradient leaves it (see _local)
            return _MaskedMeanCE.apply(logits, answers, mask, fix if self.dp_opt is None else None)
        return self.criterion(logits, answers)

    def _local(self, inputs, answers, mask=None, fix=None, with_main_step=True):
        self._zero(inputs)
        # A padded step divides the loss by the padded row count.  The non-private gradients become those
        # of the mean over the real rows again through the factor padded / real (a device scalar), applied
        # to the gradient that leaves the private module (its own per-sample gradients keep the padded
        # scale, which the DP step undoes with a host-side factor that a captured graph cannot vary).
        if self.dp_opt is not None and self.main_opt is not None:
            self.dp_opt.module.input_grad_scale = fix
        try:
            logits = self.forward(*inputs)
            loss = self._loss(logits, answers, mask, fix)
            loss.backward(gradient=ops.unit_gradient(loss))
        finally:
            if self.dp_opt is not None:
                self.dp_opt.module.input_grad_scale = None
        if with_main_step and self.main_opt is not None:
            self.main_opt.step()
        if self.dp_opt is not None:
            self.dp_opt.local_step()
        # detached: a caller that keeps the loss of an eager step alive would keep its autograd graph
        # alive with it, AccumulateGrad nodes bound to the stream of that step included; a later
        # capture (on its own stream) then runs them on the old stream and the runtime crashes when
        # the capture ends
        return loss.detach(), logits.detach()

    def _exchange(self, n_real, grads=None):
        if self.reducer is not None:
            self.reducer.reduce(n_real, grads=grads)
        if self.dp_opt is not None:
            self.dp_opt.reduce_step()

    def _apply(self, with_main_step):
        if with_main_step and self.main_opt is not None:
            self.main_opt.step()
        if self.dp_opt is not None:
            self.dp_opt.apply_step()

    def eager_step(self, inputs, answers):
        """The same step launched kernel by kernel (what every signature does until it is captured)."""
        multi = self.world > 1
        loss, logits = self._local(inputs, answers, with_main_step=not multi)
        if multi:
            self._exchange(int(answers.shape[0]))
        self._apply(with_main_step=multi)
        return loss, logits

    # -- capture and replay -----------------------------------------------------------------------
    def _signature(self, inputs, answers, padded):
        if self.resident:
            return tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in (*inputs, answers))
        return tuple(((padded,) + tuple(t.shape[1:]), t.dtype) for t in (*inputs, answers))

    def _capture(self, sig, inputs, answers, padded):
        multi = self.world > 1
        dev = answers.device
        st = {"mask": None, "fix": None}
        if self.resident:
            st["inputs"], st["answers"] = list(inputs), answers
        else:
            def room(t):
                buf = torch.zeros((padded,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                buf[:t.shape[0]].copy_(t)
                return buf.requires_grad_(t.requires_grad)
            st["inputs"], st["answers"] = [room(t) for t in inputs], room(answers)
        if self.pad_to:
            st["mask"] = torch.zeros(padded, dtype=torch.float32, device=dev)
            st["mask"][:answers.shape[0]] = 1.0
            st["fix"] = torch.full((), padded / max(1, answers.shape[0]), dtype=torch.float32, device=dev)
        hook = None
        if self.dp_opt is not None:  # the accountant counts replays, not the capture
            hook, self.dp_opt.accountant_hook = self.dp_opt.accountant_hook, None
        mode = "thread_local" if multi else "global"
        try:
            torch.cuda.synchronize()
            g_local = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_local, pool=self._pool, capture_error_mode=mode):
                loss, logits = self._local(st["inputs"], st["answers"], st["mask"], st["fix"], with_main_step=not multi)
                if not multi:
                    self._apply(with_main_step=False)
            self._pool = g_local.pool()
            st.update(local=g_local, loss=loss, logits=logits, apply=None, grads=None)
            if multi:
                if self.reducer is not None:
                    st["grads"] = [p.grad for p in self.reducer.params]
                g_local.replay()
                self._exchange(int(answers.shape[0]), st["grads"])
                g_apply = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_apply, pool=self._pool, capture_error_mode=mode):
                    self._apply(with_main_step=True)
                st["apply"] = g_apply
                g_apply.replay()
                st["ran"] = True  # the capture sequence above executed this step
            ok = 1
        except Exception as exc:  # capture is an optimisation of the launch path only
            import sys
            print(f"[vqa_amd] HIP graph capture unavailable for this step ({type(exc).__name__}: {exc}); "
                  "launching eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            ok, st = 0, None
        finally:
            if self.dp_opt is not None:
                self.dp_opt.accountant_hook = hook
        if multi:
            # every rank takes the same path: the exchange pattern of a replayed step differs from the
            # eager one only in launch mechanics, but a uniform run keeps `launch` true for all ranks
            import torch.distributed as dist
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                st = None
        return st

    def __call__(self, inputs, answers):
        """Runs one step; returns (loss, logits) -- after a replay these are the graph's own output
        tensors, overwritten by the next step of the same signature (logits include padded rows)."""
        inputs = tuple(inputs)
        n_real = int(answers.shape[0])
        padded = n_real
        if self.pad_to and not self.resident:
            padded = (n_real + self.pad_to - 1) // self.pad_to * self.pad_to
        sig = self._signature(inputs, answers, padded)
        ent = self._sigs.get(sig)
        if ent is None:
            ent = self._sigs[sig] = {"seen": 0, "graph": None, "eager": len(self._sigs) >= self.max_graphs}
        if ent["graph"] is None and not ent["eager"] and ent["seen"] >= self.warmup and answers.is_cuda:
            ent["graph"] = self._capture(sig, inputs, answers, padded)
            ent["eager"] = ent["graph"] is None
            if ent["graph"] is not None:
                self.launch = "hipgraph"
                if not ent["graph"].pop("ran", False):
                    return self._replay(ent["graph"], None, None, n_real, first=True)
                if self.dp_opt is not None and self.dp_opt.accountant_hook is not None:
                    self.dp_opt.accountant_hook(self.dp_opt)
                return self._result(ent["graph"], n_real)
        ent["seen"] += 1
        if ent["graph"] is None:
            return self.eager_step(inputs, answers)
        return self._replay(ent["graph"], inputs, answers, n_real)

    def _result(self, st, n_real):
        loss = st["loss"] * st["fix"] if st["fix"] is not None else st["loss"]
        return loss, st["logits"][:n_real]

    def _replay(self, st, inputs, answers, n_real, first=False):
        if not self.resident and not first:
            for buf, t in zip(st["inputs"], inputs):
                buf.data[:n_real].copy_(t, non_blocking=True)
                if n_real < buf.shape[0]:
                    buf.data[n_real:].zero_()
            st["answers"][:n_real].copy_(answers, non_blocking=True)
            if st["mask"] is not None:
                st["mask"][:n_real] = 1.0
                st["mask"][n_real:] = 0.0
                st["fix"].fill_(st["mask"].shape[0] / max(1, n_real))
        st["local"].replay()
        if st["apply"] is not None:
            self._exchange(n_real, st["grads"])
            st["apply"].replay()
        if self.dp_opt is not None and self.dp_opt.accountant_hook is not None:
            self.dp_opt.accountant_hook(self.dp_opt)
        return self._result(st, n_real)


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
