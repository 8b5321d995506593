"""DP-SGD engine for the classifier, shaped like the Opacus surface the reference uses.

The reference delegates DP-SGD to Opacus 1.4.0 (train.py:12-13, 124-153; utils/src.py:99,104):

    model.classifier = ModuleValidator.fix(model.classifier)
    privacy_engine = PrivacyEngine(secure_mode=False)
    module, optimizer, loader = privacy_engine.make_private(module=, optimizer=, data_loader=,
                                    noise_multiplier=, max_grad_norm=, poisson_sampling=True)
    ... loss.backward(); optimizer.step(); privacy_engine.get_epsilon(delta)

The classes below keep those names and call shapes.  What differs is how the work is done:
per-sample gradients are never materialised (Opacus stores a (B, out, in) tensor per Linear).
Each wrapped layer records its input a_i and its output gradient g_i; the optimizer step then
runs, on the HIP kernels of libvqafuse,

    ghost norms    n_i^2 = sum_layers ||a_i||^2 ||g_i||^2 (+ bias / affine terms)
    clip           c_i  = min(1, C / (n_i + 1e-6))
    clipped sums   S    = sum_i c_i g_i a_i^T            (batch-contraction GEMM on MFMA)
    noise + scale  grad = (S + N(0, (sigma C)^2)) / expected_batch_size

and hands `grad` to the wrapped torch optimizer.  With torch.distributed initialised the
clipped sums are all-reduced (RCCL) in one flat bucket and each rank adds an independent
N(0, (sigma C)^2 / world) share from its own rank-seeded stream, so the total noise is added
exactly once.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import check, dtype_code, load, ptr, stream_ptr
from .accountant import RDPAccountant, get_noise_multiplier

# fp32 tapes: the Linear layers' clipped weight sums S_W = G^T A run as library GEMMs (True) or as problems of
# this library's grouped f32 launch (False); see DPOptimizer.clip_and_accumulate.
F32_SUMS_IN_LIBRARY = True


# --------------------------------------------------------------------------------------------
# ModuleValidator.fix  (train.py:124)
# --------------------------------------------------------------------------------------------
class ModuleValidator:
    """BatchNorm mixes samples, which per-sample clipping cannot allow; `fix` swaps every
    BatchNorm1d(n) for GroupNorm(min(32, n), n, affine) as Opacus' fixer does (fresh affine
    parameters), in place for containers, and returns the module."""

    SUPPORTED = (nn.Linear, nn.GroupNorm)

    @staticmethod
    def fix(module: nn.Module) -> nn.Module:
        if isinstance(module, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            return ModuleValidator._to_groupnorm(module)
        for name, child in list(module.named_children()):
            setattr(module, name, ModuleValidator.fix(child))
        return module

    @staticmethod
    def _to_groupnorm(bn: nn.Module) -> nn.GroupNorm:
        gn = nn.GroupNorm(min(32, bn.num_features), bn.num_features, affine=bn.affine)
        ref = next(bn.parameters(), None)
        if ref is not None:
            gn = gn.to(device=ref.device, dtype=ref.dtype)
        return gn

    @staticmethod
    def validate(module: nn.Module, strict: bool = False) -> List[str]:
        errors = []
        for name, m in module.named_modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
                errors.append(f"{name}: BatchNorm is not DP-compatible (use ModuleValidator.fix)")
            elif next(m.parameters(recurse=False), None) is not None and \
                    not isinstance(m, ModuleValidator.SUPPORTED):
                errors.append(f"{name}: {type(m).__name__} has parameters but no per-sample rule")
        if strict and errors:
            raise ValueError("; ".join(errors))
        return errors

    @staticmethod
    def is_valid(module: nn.Module) -> bool:
        return not ModuleValidator.validate(module)


# --------------------------------------------------------------------------------------------
# Layer wrappers that record (a_i, g_i) instead of computing parameter gradients
# --------------------------------------------------------------------------------------------
class _Tape:
    """What one wrapped layer leaves behind for the optimizer step."""

    __slots__ = ("module", "kind", "a", "g", "shadow")

    def __init__(self, module: nn.Module, kind: str, shadow=None):
        self.module, self.kind, self.a, self.g, self.shadow = module, kind, None, None, shadow


def _as_dtype(p: torch.Tensor, dtype: torch.dtype, shadow) -> torch.Tensor:
    """The parameter in the activation dtype: as is, from the optimizer's bf16 copy (kept current
    by the fused step, so no cast kernel runs), or by a cast."""
    if p.dtype == dtype:
        return p
    if shadow is not None and dtype == torch.bfloat16:
        v = shadow.get(p.data_ptr())
        if v is not None:
            return v
    return p.to(dtype)


class _GnLink:
    """What a fused Linear -> GroupNorm block leaves for the Linear that consumes its output: that
    Linear's backward can run the norm's backward as the epilogue of its input-gradient GEMM
    (vqaf_linear_dgrad_groupnorm_bwd) and hands back the gradient at the norm's INPUT; `done` tells the
    block's own backward that its incoming gradient is already that one."""

    __slots__ = ("y", "xhat", "rstd", "ga", "groups", "relu", "out_scale", "gn_tape", "done")

    def __init__(self):
        self.y = self.xhat = self.rstd = self.ga = self.gn_tape = None
        self.groups, self.relu, self.out_scale, self.done = 0, 0, 1.0, False


def _dgrad_with_norm_backward(link, gy, w, Bn, In, Out):
    """gx = GroupNorm_backward(gy W) in one launch when `link` describes the norm in front and the shapes are
    covered; None otherwise.  Fills the norm layer's tape."""
    if link is None or link.xhat is None or gy.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or \
            not w.is_contiguous() or link.xhat.shape != (Bn, In):
        return None
    if not load().vqaf_linear_dgrad_groupnorm_supported(gy.stride(0), In, Bn, In, Out, link.groups):
        return None
    g_pre = torch.empty_like(link.xhat)
    gl = torch.empty_like(link.xhat)
    check(load().vqaf_linear_dgrad_groupnorm_bwd(
        gy.data_ptr(), gy.stride(0), ptr(w), In, Bn, In, Out, ptr(link.y) if link.relu else None, ptr(link.xhat),
        ptr(link.rstd), ptr(link.ga), dtype_code(link.ga), link.groups, int(link.relu), float(link.out_scale),
        ptr(g_pre), ptr(gl), stream_ptr(gy.device)), "vqaf_linear_dgrad_groupnorm_bwd")
    link.gn_tape.a, link.gn_tape.g = link.xhat, g_pre
    link.done = True
    return gl


class _GhostLinear(torch.autograd.Function):
    """y = x W^T + b through the vendor GEMM; backward returns only dL/dx and records dL/dy.
    The weight gradient is NOT computed here -- the clipped sum replaces it."""

    @staticmethod
    def forward(ctx, x, weight, bias, tape, link=None):
        w = _as_dtype(weight, x.dtype, tape.shadow)
        b = None if bias is None else _as_dtype(bias, x.dtype, tape.shadow)
        Bn, In = x.shape
        Out = w.shape[0]
        xc = x.contiguous()
        if x.dtype == torch.bfloat16 and w.is_contiguous() and \
                ops.gemm_bf16_supported(In, False, In, False, Bn, Out, In):
            # y = x W^T + b on the library's own bf16 MFMA GEMM (bias in the epilogue)
            y = torch.empty(Bn, Out, dtype=torch.bfloat16, device=x.device)
            ops.gemm_bf16(xc, False, w, False, Bn, Out, In, y, bias=b)
        else:
            y = F.linear(x, w, b)
        ctx.save_for_backward(w)
        ctx.tape = tape
        ctx.link = link
        tape.a = xc.detach()
        return y

    @staticmethod
    def backward(ctx, gy):
        (w,) = ctx.saved_tensors
        if ops.zero_padded_pitch(gy) is None:  # (a zero-padded backprop of the loss is used where it lies)
            gy = gy.contiguous()
        ctx.tape.g = gy
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        Bn, Out = gy.shape
        In = w.shape[1]
        gl = _dgrad_with_norm_backward(ctx.link, gy, w, Bn, In, Out)
        if gl is not None:
            return gl, None, None, None, None
        if gy.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.is_contiguous() and \
                ops.gemm_bf16_supported(gy.stride(0), False, In, True, Bn, In, Out):
            gx = torch.empty(Bn, In, dtype=torch.bfloat16, device=gy.device)
            ops.gemm_bf16(gy, False, w, True, Bn, In, Out, gx)
        else:
            gx = gy.matmul(w)
        return gx, None, None, None, None


class _GhostGroupNorm(torch.autograd.Function):
    """GroupNorm (+ fused ReLU) on (B,C) by vqaf_groupnorm_fwd/bwd; records xhat and the
    gradient at the affine output."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu, tape, drop_p=0.0, drop_seed=0, drop_counter=None):
        B, C = x.shape
        xc = x.contiguous()
        ga = _as_dtype(weight.detach(), x.dtype, tape.shadow).contiguous()
        be = _as_dtype(bias.detach(), x.dtype, tape.shadow).contiguous()
        y = torch.empty_like(xc)
        xhat = torch.empty_like(xc)
        rstd = torch.empty(B, groups, dtype=torch.float32, device=x.device)
        check(load().vqaf_groupnorm_fwd(ptr(xc), ptr(ga), ptr(be), B, C, groups, float(eps), int(relu),
                                        dtype_code(xc), ptr(y), ptr(xhat), ptr(rstd), float(drop_p),
                                        int(drop_seed) & 0xFFFFFFFFFFFFFFFF,
                                        ptr(drop_counter) if drop_p > 0 else None,
                                        stream_ptr(x.device)), "vqaf_groupnorm_fwd")
        ctx.save_for_backward(y, xhat, rstd, ga)
        ctx.meta = (B, C, groups, int(relu), 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0)
        ctx.tape = tape
        return y

    @staticmethod
    def backward(ctx, gy):
        y, xhat, rstd, ga = ctx.saved_tensors
        B, C, groups, relu, out_scale = ctx.meta
        gy = gy.contiguous().to(y.dtype)
        g_pre = torch.empty_like(y)
        gx = torch.empty_like(y)
        check(load().vqaf_groupnorm_bwd(ptr(gy), ptr(y), ptr(xhat), ptr(rstd), ptr(ga), B, C, groups,
                                        relu, float(out_scale), dtype_code(y), ptr(g_pre), ptr(gx),
                                        stream_ptr(y.device)), "vqaf_groupnorm_bwd")
        ctx.tape.a = xhat
        ctx.tape.g = g_pre
        return gx, None, None, None, None, None, None, None, None, None


class _GhostLinearGroupNorm(torch.autograd.Function):
    """Linear -> GroupNorm (-> ReLU -> Dropout) with the normalisation as the epilogue of the Linear's GEMM
    (vqaf_linear_groupnorm_fwd: one launch, the same bits as _GhostLinear followed by _GhostGroupNorm).
    Backward: the norm's backward has usually run already, as the epilogue of the consuming Linear's
    input-gradient GEMM (link_out.done; see _GnLink), otherwise the GroupNorm backward kernel runs here; then
    the Linear's input gradient, itself with the backward of the norm in FRONT of this block as its epilogue
    when there is one (link_in).  Both layers' tapes are filled as the separate functions fill them."""

    @staticmethod
    def forward(ctx, x, weight, bias, gn_weight, gn_bias, groups, eps, relu, lin_tape, gn_tape, drop_p=0.0,
                drop_seed=0, drop_counter=None, link_in=None, link_out=None):
        w = _as_dtype(weight, x.dtype, lin_tape.shadow)
        b = None if bias is None else _as_dtype(bias, x.dtype, lin_tape.shadow)
        ga = _as_dtype(gn_weight.detach(), x.dtype, gn_tape.shadow).contiguous()
        be = _as_dtype(gn_bias.detach(), x.dtype, gn_tape.shadow).contiguous()
        Bn, In = x.shape
        C = w.shape[0]
        xc = x.contiguous()
        y = torch.empty(Bn, C, dtype=torch.bfloat16, device=x.device)
        xhat = torch.empty_like(y)
        rstd = torch.empty(Bn, groups, dtype=torch.float32, device=x.device)
        check(load().vqaf_linear_groupnorm_fwd(
            ptr(xc), In, ptr(w), In, ptr(b), dtype_code(b) if b is not None else 0, Bn, C, In, ptr(ga), ptr(be),
            dtype_code(ga), groups, float(eps), int(relu), float(drop_p), int(drop_seed) & 0xFFFFFFFFFFFFFFFF,
            ptr(drop_counter) if drop_p > 0 else None, ptr(y), ptr(xhat), ptr(rstd), stream_ptr(x.device)),
            "vqaf_linear_groupnorm_fwd")
        ctx.save_for_backward(w, y, xhat, rstd, ga)
        ctx.meta = (Bn, C, groups, int(relu), 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0)
        ctx.tapes = (lin_tape, gn_tape)
        ctx.links = (link_in, link_out)
        if link_out is not None:
            # (y detached: the link must not reach this node through its own output, or the step's autograd
            # graph would stay alive until the cycle collector runs)
            link_out.y, link_out.xhat, link_out.rstd, link_out.ga = y.detach(), xhat, rstd, ga
            link_out.groups, link_out.relu, link_out.out_scale = groups, int(relu), ctx.meta[4]
            link_out.gn_tape, link_out.done = gn_tape, False
        lin_tape.a = xc.detach()
        return y

    @staticmethod
    def backward(ctx, gy):
        w, y, xhat, rstd, ga = ctx.saved_tensors
        Bn, C, groups, relu, out_scale = ctx.meta
        lin_tape, gn_tape = ctx.tapes
        link_in, link_out = ctx.links
        if link_out is not None and link_out.done:
            # the consuming Linear ran this norm's backward on its input-gradient tile: gy IS the gradient
            # at this block's Linear output, and the norm's tape is filled
            gl = gy
        else:
            gy = gy.contiguous().to(y.dtype)
            g_pre = torch.empty_like(y)
            gl = torch.empty_like(y)  # gradient at the Linear's output
            check(load().vqaf_groupnorm_bwd(ptr(gy), ptr(y), ptr(xhat), ptr(rstd), ptr(ga), Bn, C, groups, relu,
                                            float(out_scale), dtype_code(y), ptr(g_pre), ptr(gl),
                                            stream_ptr(y.device)), "vqaf_groupnorm_bwd")
            gn_tape.a, gn_tape.g = xhat, g_pre
        lin_tape.g = gl
        none = (None,) * 14
        if not ctx.needs_input_grad[0]:
            return (None,) + none
        In = w.shape[1]
        gin = _dgrad_with_norm_backward(link_in, gl, w, Bn, In, C)
        if gin is not None:
            return (gin,) + none
        if ops.gemm_bf16_supported(C, False, In, True, Bn, In, C):
            gx = torch.empty(Bn, In, dtype=torch.bfloat16, device=gy.device)
            ops.gemm_bf16(gl, False, w, True, Bn, In, C, gx)
        else:
            gx = gl.matmul(w)
        return (gx,) + none


class _ScaleGrad(torch.autograd.Function):
    """Identity whose backward multiplies the gradient by a 0-d device tensor (read on the device)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(scale)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return g * scale.to(g.dtype), None


class GradSampleModule(nn.Module):
    """Wraps the classifier; the wrapped module is `_module`, so state_dict keys gain the
    `_module.` prefix the reference's loader strips (utils/util.py:117-123)."""

    def __init__(self, m: nn.Module, *, batch_first: bool = True, loss_reduction: str = "mean"):
        super().__init__()
        errors = ModuleValidator.validate(m)
        if errors:
            raise ValueError("module is not DP-compatible: " + "; ".join(errors))
        if loss_reduction not in ("mean", "sum"):
            raise ValueError("loss_reduction must be 'mean' or 'sum'")
        self._module = m
        self.loss_reduction = loss_reduction
        self.tapes: List[_Tape] = []
        self.last_batch_size = 0
        self.shadow = None  # data_ptr of an f32 parameter -> its bf16 copy (set by DPOptimizer)
        self.dropout_rng = None  # seed (and fallback counter) of the fused dropout masks (made on first use)
        self.step_counter = None  # {unused, steps} device words of the attached DP optimizer's fused update
        self._drop_calls = 0      # dropout call sites since the last optimizer step
        # Optional 0-d device tensor: the gradient that leaves this module towards the non-private part of
        # the model is multiplied by it.  A step over a zero-padded batch divides the loss by the PADDED
        # row count (the host-side factor that restores per-sample gradients is fixed in a captured graph);
        # this factor, padded / real rows, turns the upstream gradients back into those of the mean over
        # the real rows.  The per-sample gradients of this module are not touched by it.
        self.input_grad_scale = None
        # Linear followed by GroupNorm (+ReLU +Dropout) as ONE launch where the shapes allow it
        self.fuse_linear_groupnorm = True

    def forward(self, x):
        self.tapes = []
        self._link = None
        self.last_batch_size = x.shape[0]
        if self.input_grad_scale is not None and x.requires_grad:
            x = _ScaleGrad.apply(x, self.input_grad_scale)
        return self._run(self._module, x)

    def _run(self, m: nn.Module, x):
        if isinstance(m, nn.Sequential):
            children = list(m.children())
            i = 0
            while i < len(children):
                c = children[i]
                if self.fuse_linear_groupnorm and isinstance(c, nn.Linear) and i + 1 < len(children) and \
                        isinstance(children[i + 1], nn.GroupNorm) and x.dim() == 2:
                    # Linear -> GroupNorm (-> ReLU (-> Dropout)): the normalisation rides on the GEMM's epilogue
                    gn = children[i + 1]
                    relu = i + 2 < len(children) and isinstance(children[i + 2], nn.ReLU)
                    drop = children[i + 3] if relu and i + 3 < len(children) and \
                        isinstance(children[i + 3], nn.Dropout) else None
                    p = 0.0
                    if drop is not None:
                        if drop.training and drop.p >= 1.0:
                            drop = None  # p = 1 zeroes everything: leave it to nn.Dropout
                        else:
                            p = float(drop.p) if (drop.training and 0.0 < drop.p < 1.0) else 0.0
                    if self._linear_groupnorm_ok(c, gn, x):
                        link_in, self._link = self._link, _GnLink()
                        x = self._linear_groupnorm(c, gn, x, relu=relu, drop_p=p, link_in=link_in, link_out=self._link)
                        i += 2 + (1 if relu else 0) + (1 if drop is not None else 0)
                        continue
                if isinstance(c, nn.GroupNorm) and i + 1 < len(children) and \
                        isinstance(children[i + 1], nn.ReLU) and x.dim() == 2:
                    # GroupNorm -> ReLU (-> Dropout) in one launch (model/model.py:56-58)
                    drop = children[i + 2] if i + 2 < len(children) and \
                        isinstance(children[i + 2], nn.Dropout) else None
                    if drop is not None:
                        p = float(drop.p) if (drop.training and 0.0 < drop.p < 1.0) else 0.0
                        if drop.training and drop.p >= 1.0:
                            drop = None  # p = 1 zeroes everything: leave it to nn.Dropout
                    if drop is not None:
                        x = self._groupnorm(c, x, relu=True, drop_p=p)
                        i += 3
                    else:
                        x = self._groupnorm(c, x, relu=True)
                        i += 2
                    continue
                x = self._run(c, x)
                i += 1
            return x
        link, self._link = self._link, None  # only the module that directly consumes a fused block's output may use it
        if isinstance(m, nn.Linear):
            if x.dim() != 2:
                raise ValueError("per-sample rule implemented for Linear on (B, features) input")
            tape = _Tape(m, "linear", self.shadow)
            self.tapes.append(tape)
            return _GhostLinear.apply(x, m.weight, m.bias, tape, link)
        if isinstance(m, nn.GroupNorm):
            return self._groupnorm(m, x, relu=False)
        if next(m.parameters(), None) is None:
            return m(x)  # ReLU, Dropout, Identity ...: parameter-free, plain autograd
        raise ValueError(f"{type(m).__name__} has parameters but no per-sample rule")

    def _linear_groupnorm_ok(self, lin: nn.Linear, gn: nn.GroupNorm, x) -> bool:
        if x.dtype != torch.bfloat16 or not gn.affine or gn.num_channels != lin.out_features:
            return False
        return bool(load().vqaf_linear_groupnorm_supported(lin.in_features, lin.in_features, x.shape[0],
                                                           lin.out_features, lin.in_features, gn.num_groups))

    def _dropout_stream(self, x, drop_p: float):
        """(seed, counter tensor) of the next dropout call site.  The mask stream is (seed + call site,
        step count).  With a DP optimizer attached the count is the one its fused update keeps in device
        memory (it moves once per step, so no launch is spent on a counter here and a replayed HIP graph
        draws fresh masks); the call-site index tells apart the dropout layers and repeated forwards of
        one step.  Without one the module's own counter is used and bumped after each forward."""
        if drop_p <= 0.0:
            return 0, None
        if self.dropout_rng is None or self.dropout_rng.state.device != x.device:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            # seeded from torch's global seed; ranks that share it still draw different masks
            self.dropout_rng = ops.PhiloxState(x.device, stream_salt=29 + 1009 * rank)
        counter = self.step_counter if self.step_counter is not None else self.dropout_rng.state
        seed = self.dropout_rng.seed + 0x9E3779B97F4A7C15 * (1 + self._drop_calls)
        self._drop_calls += 1
        return seed, counter

    def _linear_groupnorm(self, lin: nn.Linear, gn: nn.GroupNorm, x, relu: bool, drop_p: float = 0.0,
                          link_in=None, link_out=None):
        lin_tape = _Tape(lin, "linear", self.shadow)
        gn_tape = _Tape(gn, "affine", self.shadow)
        self.tapes += [lin_tape, gn_tape]
        seed, counter = self._dropout_stream(x, drop_p)
        y = _GhostLinearGroupNorm.apply(x, lin.weight, lin.bias, gn.weight, gn.bias, gn.num_groups, gn.eps, relu,
                                        lin_tape, gn_tape, drop_p, seed, counter, link_in, link_out)
        if drop_p > 0.0 and self.step_counter is None:
            self.dropout_rng.advance()
        return y

    def _groupnorm(self, m: nn.GroupNorm, x, relu: bool, drop_p: float = 0.0):
        if x.dim() != 2 or not m.affine:
            raise ValueError("per-sample rule implemented for affine GroupNorm on (B, C) input")
        self._link = None  # a stand-alone norm consumes the previous block's output itself
        tape = _Tape(m, "affine", self.shadow)
        self.tapes.append(tape)
        seed, counter = self._dropout_stream(x, drop_p)
        y = _GhostGroupNorm.apply(x, m.weight, m.bias, m.num_groups, m.eps, relu, tape, drop_p, seed, counter)
        if drop_p > 0.0 and self.step_counter is None:
            self.dropout_rng.advance()
        return y


# --------------------------------------------------------------------------------------------
# DPOptimizer
# --------------------------------------------------------------------------------------------
class DPOptimizer:
    """clip -> sum -> (all-reduce) -> noise -> scale -> wrapped optimizer step."""

    def __init__(self, optimizer: torch.optim.Optimizer, module: GradSampleModule, *,
                 noise_multiplier: float, max_grad_norm: float, expected_batch_size: int,
                 seed: Optional[int] = None, process_group=None, fuse_adamw: bool = True):
        self.original_optimizer = optimizer
        self.fuse_adamw = bool(fuse_adamw)
        self.module = module
        self.noise_multiplier = float(noise_multiplier)
        self.max_grad_norm = float(max_grad_norm)
        self.expected_batch_size = int(expected_batch_size)
        self.process_group = process_group
        self.accountant_hook = None
        self.params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        self._offsets = {}
        off = 0
        for p in self.params:
            self._offsets[id(p)] = off
            off += (p.numel() + 3) // 4 * 4  # keep every slice 16-byte aligned
        self._numel = off
        self._S = None
        self._grad = None
        self._rng = None
        self._seed = seed
        self._P = self._M = self._V = self._shadow = self._step = None
        self._pending_state = None  # optimizer state loaded before the fused buckets exist
        self._use_fused = False
        # With one rank, in-kernel noise and one parameter group, the clipped-sum launch of the Linear layers
        # can apply the update itself (vqaf_clipped_sum_adamw_group); _updated lists what it covered this
        # step.  Off by default: the update's 130 MB of parameter and moment traffic is not hidden under the
        # GEMM (56 + 7 us against 31 + 30 us for the two launches at batch 1024), so nothing is gained yet.
        self.fuse_update_into_sums = False
        self._updated = None
        self._step_counted = False
        self.last_norms = None
        self._last_coef, self._last_scale = None, 1.0

    @property
    def last_clip(self):
        """Per-sample clip factors min(1, C / (n_i + 1e-6)) of the last step (computed on demand)."""
        return None if self._last_coef is None else self._last_coef / self._last_scale

    # -- torch.optim.Optimizer surface the training loop touches ------------------------------
    @property
    def param_groups(self):
        return self.original_optimizer.param_groups

    def state_dict(self):
        if self._step is not None:  # publish the fused step's moments in torch's own layout
            t = float(self._step[1].item())
            for p in self.params:
                o = self._offsets[id(p)]
                self.original_optimizer.state[p] = {
                    "step": torch.tensor(t), "exp_avg": self._M[o:o + p.numel()].view_as(p),
                    "exp_avg_sq": self._V[o:o + p.numel()].view_as(p)}
        return self.original_optimizer.state_dict()

    def load_state_dict(self, sd):
        """Restores the wrapped optimizer's state; when the fused step is (or will be) in use the
        loaded moments and step count are copied into its flat buckets, so a resumed run continues
        from them instead of from zero."""
        self.original_optimizer.load_state_dict(sd)
        state = self.original_optimizer.state
        if not any(len(state.get(p, {})) for p in self.params):
            return
        self._pending_state = {id(p): dict(state[p]) for p in self.params if len(state.get(p, {}))}
        if self._P is not None:
            self._adopt_loaded_state()

    @torch.no_grad()
    def _adopt_loaded_state(self):
        pend, self._pending_state = self._pending_state, None
        if not pend:
            return
        step = 0
        for p in self.params:
            st = pend.get(id(p))
            if st is None:
                continue
            o = self._offsets[id(p)]
            self._M[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self._V[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(st["step"])))
        self._step[1] = step
        # the fused kernel owns the state from here on (state_dict() publishes it back)
        for p in self.params:
            self.original_optimizer.state.pop(p, None)

    def zero_grad(self, set_to_none: bool = False):
        # Parameter .grad tensors are views of the flat buffer and are overwritten by every
        # step; the only per-step state is the recorded backprops.
        for t in self.module.tapes:
            t.g = None

    def attach_step_hook(self, fn):
        self.accountant_hook = fn

    # -- the DP step ---------------------------------------------------------------------------
    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
        return 1, 0

    def _buffers(self, device):
        if self._S is None or self._S.device != device:
            self._S = torch.zeros(self._numel, dtype=torch.float32, device=device)
            self._grad = torch.zeros(self._numel, dtype=torch.float32, device=device)
            for p in self.params:
                o = self._offsets[id(p)]
                p.grad = self._grad[o:o + p.numel()].view_as(p)
            world, rank = self._world()
            seed = self._seed if self._seed is not None else torch.initial_seed()
            self._rng = ops.PhiloxState(device, seed=seed, stream_salt=1 + rank)
        else:
            # an outside model.zero_grad(set_to_none=True) drops the views; without them the
            # wrapped optimizer would silently skip those parameters
            for p in self.params:
                if p.grad is None:
                    o = self._offsets[id(p)]
                    p.grad = self._grad[o:o + p.numel()].view_as(p)
        return self._S, self._grad

    def _fusable(self) -> bool:
        """True when the wrapped optimizer is a plain torch AdamW over f32 parameters, whose update
        the fused noise+scale+AdamW kernel reproduces exactly."""
        opt = self.original_optimizer
        if not self.fuse_adamw or type(opt) is not torch.optim.AdamW:
            return False
        if any(g.get("amsgrad") or g.get("maximize") or isinstance(g["lr"], torch.Tensor)
               for g in opt.param_groups):
            return False
        if not all(p.dtype == torch.float32 for p in self.params):
            return False
        # a state loaded before the first step is adopted by the fused buckets; any other
        # pre-existing state (steps already taken by the wrapped optimizer itself) is not
        return len(opt.state) == 0 or self._pending_state is not None

    def _fused_buffers(self, device):
        """Flat parameter / moment buckets; every Parameter becomes a view of the flat bucket
        (so move the module to its device BEFORE the first step; a later .to() would detach the
        parameters from the bucket)."""
        if self._P is None:
            self._P = torch.zeros(self._numel, dtype=torch.float32, device=device)
            self._M = torch.zeros_like(self._P)
            self._V = torch.zeros_like(self._P)
            self._shadow = torch.zeros(self._numel, dtype=torch.bfloat16, device=device)
            # {seed, updates applied}: the update count doubles as the counter of the noise stream,
            # so one advance per step serves both the bias correction and the noise
            self._step = ops.PhiloxState(device, seed=self._rng.seed, stream_salt=17).state
            lookup = {}
            with torch.no_grad():
                for p in self.params:
                    o = self._offsets[id(p)]
                    view = self._P[o:o + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                    lookup[view.data_ptr()] = self._shadow[o:o + p.numel()].view_as(p)
                self._shadow.copy_(self._P)
            self.module.shadow = lookup
            self._adopt_loaded_state()

    @torch.no_grad()
    def _fused_apply(self, noise: Optional[torch.Tensor] = None):
        """noise + scale + AdamW + bf16 refresh in one pass per parameter group.  With one rank the
        kernel reads the clipped sums and adds the noise itself; in a process group it reads the
        already noised, scaled and all-reduced gradient bucket."""
        S, grad = self._S, self._grad
        dev = S.device
        self._fused_buffers(dev)
        lib, st = load(), stream_ptr(dev)
        world, _ = self._world()
        if world == 1:
            src, std, scale = S, self.noise_multiplier * self.max_grad_norm, 1.0 / self.expected_batch_size
        else:
            src, std, scale, noise = grad, 0.0, 1.0, None
        single = len(self.original_optimizer.param_groups) == 1
        counted, self._step_counted = self._step_counted, False  # the clipping launch already counted this step
        if self._updated:
            # the clipped-sum launch already updated the Linear layers: the rest (normalisation layers) in
            # one launch over their ranges, with the noise addressing of the whole-bucket launch
            if noise is not None:
                raise RuntimeError("the update was fused into the clipped sums (in-kernel noise); call "
                                   "clip_and_accumulate() without fuse_update to inject noise")
            import ctypes
            group, lo = self._single_group()
            b1, b2 = group["betas"]
            rest = [p for p in group["params"] if id(p) in self._offsets and id(p) not in self._updated]
            for i in range(0, len(rest), 8):
                part = rest[i:i + 8]
                check(lib.vqaf_dp_adamw_ranges(
                    len(part), (ctypes.c_long * len(part))(*[self._offsets[id(p)] for p in part]),
                    (ctypes.c_size_t * len(part))(*[p.numel() for p in part]), ptr(S), ptr(self._step), lo, std, scale,
                    ptr(self._P), ptr(self._M), ptr(self._V), ptr(self._shadow), ptr(grad), float(group["lr"]),
                    float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), ptr(self._step), st),
                    "vqaf_dp_adamw_ranges")
            self._updated = None
            check(lib.vqaf_rng_advance(ptr(self._step), 1, st), "vqaf_rng_advance")
            self.module.step_counter = self._step
            self.module._drop_calls = 0
            return
        for group in self.original_optimizer.param_groups:
            ps = [p for p in group["params"] if id(p) in self._offsets]
            if not ps:
                continue
            lo = min(self._offsets[id(p)] for p in ps)
            hi = max(self._offsets[id(p)] + (p.numel() + 3) // 4 * 4 for p in ps)
            b1, b2 = group["betas"]
            nz = noise[lo:hi] if noise is not None else None
            noise_state = self._step if single else self._rng.state
            # (in a process group the kernel reads the reduced gradient bucket itself: nothing to write back)
            check(lib.vqaf_dp_adamw(ptr(src[lo:hi]), ptr(nz), ptr(noise_state), std, scale,
                                    ptr(self._P[lo:hi]), ptr(self._M[lo:hi]), ptr(self._V[lo:hi]),
                                    ptr(self._shadow[lo:hi]), ptr(grad[lo:hi]) if world == 1 else None, hi - lo,
                                    float(group["lr"]),
                                    float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                    ptr(self._step), -1 if counted else 0, st), "vqaf_dp_adamw")
            if world == 1 and nz is None and std != 0.0 and not single:
                self._rng.advance()  # several groups: a fresh stream for the next group / step
        # (step_advance = 1 would count the step in the kernel, but its returning device-scope atomic per
        # workgroup costs more than a one-thread launch at 4096 short workgroups; the clipping launch of
        # local_step() counts the step on the side instead, and only a step without one pays this launch)
        if not counted:
            check(lib.vqaf_rng_advance(ptr(self._step), 1, st), "vqaf_rng_advance")
        self.module.step_counter = self._step
        self.module._drop_calls = 0

    def _fused_step(self, noise: Optional[torch.Tensor] = None):
        """(tests) fused path with an injected noise tensor, after clip_and_accumulate()."""
        world, _ = self._world()
        self._use_fused = True
        if world > 1:
            self._noise_scale_local(noise)
            self.reduce_step()
        self._fused_apply(noise)

    def _slice(self, buf, p):
        o = self._offsets[id(p)]
        return buf[o:o + p.numel()]

    @torch.no_grad()
    def clip_and_accumulate(self, fuse_update: bool = False, count_step: bool = False):
        """Fills the flat clipped-sum buffer from the layer tapes; returns it.  With fuse_update the launch
        that forms the Linear layers' sums also applies their noise + AdamW update (those slices of the
        buffer are then not written; self._updated names the parameters).  With count_step the clipping
        launch also advances the device-side update count that the fused update reads (self._step_counted
        tells _fused_apply so): the step then spends no launch of its own on the count."""
        self._updated = None
        self._step_counted = False
        tapes = [t for t in self.module.tapes if t.g is not None]
        if not tapes:
            raise RuntimeError("DPOptimizer.step() called before loss.backward()")
        lib = load()
        dev = tapes[0].g.device
        S, _ = self._buffers(dev)
        B = tapes[0].g.shape[0]
        st = stream_ptr(dev)
        for t in tapes:
            a = t.a.contiguous()
            g = t.g if ops.zero_padded_pitch(t.g) is not None else t.g.contiguous()
            t.a = a if a.dtype == g.dtype else a.to(g.dtype)
            t.g = g
        # a mean-reduced loss hands back gradients divided by B; undo it (Opacus: loss_reduction)
        scale = float(B) if self.module.loss_reduction == "mean" else 1.0
        # bf16 tapes: the clipped rows coef * g are re-rounded to bf16 on their way into the MFMA
        # (up to 2^-9 relative per element), which could lift a clipped sample's norm just above
        # C.  Clipping to C (1 - 2^-8) keeps every sample's contribution <= C after the rounding,
        # so the sensitivity the accountant assumes holds exactly.
        max_norm = self.max_grad_norm
        if any(t.g.dtype == torch.bfloat16 for t in tapes):
            max_norm *= 1.0 - 2.0 ** -8
        coef = torch.empty(B, dtype=torch.float32, device=dev)
        norms = torch.empty(B, dtype=torch.float32, device=dev)
        n = len(tapes)
        prescaled = False
        same_dtype = all(t.g.dtype == tapes[0].g.dtype for t in tapes)

        def zero_list(prescaled_):
            # Sums that are built with atomic adds (the affine sums; a Linear whose bf16 sum is split over
            # the batch) start from zero
            out = []
            for t in tapes:
                m = t.module
                if t.kind != "linear":
                    out += [self._slice(S, m.weight), self._slice(S, m.bias)]
                elif t.g.dtype == torch.bfloat16 and not (
                        prescaled_ and t.a.dtype == torch.bfloat16 and
                        ops.gemm_bf16_supported(t.g.stride(0), True, t.a.shape[1], True, t.g.shape[1], t.a.shape[1], B)):
                    out += [self._slice(S, m.weight)] + ([self._slice(S, m.bias)] if m.bias is not None else [])
                elif prescaled_ and F32_SUMS_IN_LIBRARY and t.g.dtype == torch.float32 and m.bias is not None:
                    out += [self._slice(S, m.bias)]  # (column sums of G over the batch, split over workgroups)
            return out

        pre_zeroed = False
        if n <= 8 and same_dtype:
            import ctypes
            pa = (ctypes.c_void_p * n)(*[ptr(t.a) for t in tapes])
            # a zero-padded backprop enters with its padded width (the zeros change neither norm nor scaling)
            pg = (ctypes.c_void_p * n)(*[t.g.data_ptr() for t in tapes])
            din = (ctypes.c_int * n)(*[t.a.shape[1] for t in tapes])
            dout = (ctypes.c_int * n)(*[t.g.stride(0) for t in tapes])
            kind = (ctypes.c_int * n)(*[2 if t.kind != "linear" else int(t.module.bias is not None)
                                        for t in tapes])
            # the same launch scales every recorded backprop row by its clip coefficient in place
            # (the tapes are dead after this step), so the sums below are plain contractions
            # ... and clears the slices that the sums below build with atomic adds
            zero = zero_list(True)
            nz = len(zero) if len(zero) <= 8 else 0
            zp = (ctypes.c_void_p * max(nz, 1))(*[ptr(z) for z in zero[:nz]])
            zn = (ctypes.c_size_t * max(nz, 1))(*[z.numel() for z in zero[:nz]])
            count_step = count_step and self._step is not None and not fuse_update
            check(lib.vqaf_ghost_clip_zero_bump(n, pa, pg, din, dout, kind, B, dtype_code(tapes[0].g),
                                                max_norm, scale, 1, ptr(coef), ptr(norms), None, zp, zn, nz,
                                                ptr(self._step) if count_step else None, st),
                  "vqaf_ghost_clip_zero_bump")
            self._step_counted = count_step
            prescaled = True
            pre_zeroed = nz > 0
        else:
            sq = torch.empty(B, dtype=torch.float32, device=dev)
            for i, t in enumerate(tapes):
                t.g = t.g.contiguous()  # (a zero-padded backprop is only used in place by the one-launch path)
                a, g = t.a, t.g
                if t.kind == "linear":
                    check(lib.vqaf_ghost_sqnorm_linear(ptr(a), ptr(g), B, a.shape[1], g.shape[1],
                                                       dtype_code(g), int(t.module.bias is not None),
                                                       int(i > 0), ptr(sq), st), "vqaf_ghost_sqnorm_linear")
                else:
                    check(lib.vqaf_ghost_sqnorm_affine(ptr(a), ptr(g), B, g.shape[1], dtype_code(g),
                                                       int(i > 0), ptr(sq), st), "vqaf_ghost_sqnorm_affine")
            check(lib.vqaf_clip_coef(ptr(sq), B, max_norm, scale, ptr(coef), ptr(norms), st),
                  "vqaf_clip_coef")
        self.last_norms, self._last_coef, self._last_scale = norms, coef, scale
        if not pre_zeroed:
            # one launch zeroes all those slices, instead of a memset each
            zero = zero_list(prescaled)
            pre_zeroed = 0 < len(zero) <= 8
            if pre_zeroed:
                import ctypes
                zp = (ctypes.c_void_p * len(zero))(*[ptr(z) for z in zero])
                zn = (ctypes.c_size_t * len(zero))(*[z.numel() for z in zero])
                check(lib.vqaf_zero_multi(zp, zn, len(zero), st), "vqaf_zero_multi")
        acc = 1 if pre_zeroed else 0
        # the wide Linear layers' sums (S = G^T A, column sums of G) share ONE launch when each fills the chip
        grouped = []
        if prescaled:
            cand = [t for t in tapes if t.kind == "linear" and t.g.dtype == torch.bfloat16 and
                    t.a.dtype == torch.bfloat16 and t.g.stride(1) == 1 and
                    ops.gemm_bf16_supported(t.g.stride(0), True, t.a.shape[1], True, t.g.shape[1], t.a.shape[1], B)]
            if 2 <= len(cand) <= 4:
                probe = "clipped_sum_linear:" + "+".join(f"{t.g.shape[1]}x{t.a.shape[1]}" for t in cand)
                if fuse_update and self._sums_with_update(cand, B, probe):
                    grouped = cand
                    self._updated = {id(p) for t in cand for p in (t.module.weight, t.module.bias) if p is not None}
            if 2 <= len(cand) <= 4 and not grouped:
                items = [(t.g, t.a, self._slice(S, t.module.weight).view(t.g.shape[1], t.a.shape[1]),
                          self._slice(S, t.module.bias) if t.module.bias is not None else None) for t in cand]
                if ops.gemm_bf16_wgrad_group(items, B, name=probe):
                    grouped = cand
            # f32 tapes: S_W = G^T A is a plain f32 product of two row-major matrices -- the library GEMM (hipBLASLt
            # through torch.mm, 128-wide tiles) runs it at about 115 TFLOP/s, the 64 x 64 tiles of this library at
            # 65-80 -- and every S_b = 1^T G is a problem of ONE grouped launch.  F32_SUMS_IN_LIBRARY = False puts the
            # S_W products into that launch too, everything written with plain stores (nothing to clear beforehand).
            f32 = [t for t in tapes if t.kind == "linear" and t.g.dtype == torch.float32 and
                   t.a.dtype == torch.float32 and t.g.is_contiguous() and t.a.is_contiguous()]
            if f32 and sum(1 + (t.module.bias is not None) for t in f32) <= 12 and B >= 64:
                from .xattn import _gemm_group, _ones, _prob
                one = _ones(dev)
                probs = []
                for t in f32:
                    In_, Out_ = t.a.shape[1], t.g.shape[1]
                    if not F32_SUMS_IN_LIBRARY:
                        probs.append(_prob((t.g, 0), (0, 1, Out_), (t.a, 0), (0, In_, 1),
                                           (self._slice(S, t.module.weight), 0), (0, In_), 1, Out_, In_, B, store=True))
                    if t.module.bias is not None:
                        # a 1 x Out product: stored by one workgroup per 64 outputs it would be one long chain of
                        # fetches, so it is split over the batch and added into its slice (cleared by the clip launch)
                        probs.append(_prob((one, 0), (0, 0, 0), (t.g, 0), (0, Out_, 1),
                                           (self._slice(S, t.module.bias), 0), (0, Out_), 1, 1, Out_, B,
                                           store=not (pre_zeroed and F32_SUMS_IN_LIBRARY)))
                probe = "clipped_sum_linear:" + "+".join(f"{t.g.shape[1]}x{t.a.shape[1]}" for t in f32)
                ops._probe(probe, "begin")
                if F32_SUMS_IN_LIBRARY:
                    for t in f32:
                        torch.mm(t.g.t(), t.a, out=self._slice(S, t.module.weight).view(t.g.shape[1], t.a.shape[1]))
                if probs:
                    _gemm_group(probs)
                ops._probe(probe, "end")
                grouped = grouped + f32
        # the affine (GroupNorm) layers' sums share one launch too when their slices are already zeroed
        affine = [t for t in tapes if t.kind != "linear"]
        if pre_zeroed and 2 <= len(affine) <= 8 and all(t.g.dtype == affine[0].g.dtype and t.g.is_contiguous()
                                                         and t.a.is_contiguous() for t in affine):
            import ctypes
            n = len(affine)
            vp = ctypes.c_void_p * n
            check(lib.vqaf_clipped_sum_affine_multi(
                n, vp(*[ptr(t.a) for t in affine]), vp(*[ptr(t.g) for t in affine]), None if prescaled else ptr(coef),
                B, (ctypes.c_int * n)(*[t.g.shape[1] for t in affine]), dtype_code(affine[0].g),
                vp(*[ptr(self._slice(S, t.module.weight)) for t in affine]),
                vp(*[ptr(self._slice(S, t.module.bias)) for t in affine]), st), "vqaf_clipped_sum_affine_multi")
            grouped = grouped + affine
        for t in tapes:
            if any(t is c for c in grouped):
                continue
            a, g = t.a, (t.g if ops.zero_padded_pitch(t.g) is not None else t.g.contiguous())
            m = t.module
            if t.kind == "linear":
                sw = self._slice(S, m.weight)
                sb = self._slice(S, m.bias) if m.bias is not None else None
                probe = f"clipped_sum_linear:{g.shape[1]}x{a.shape[1]}"
                In_, Out_ = a.shape[1], g.shape[1]
                if prescaled and g.dtype == torch.bfloat16 and a.dtype == torch.bfloat16 and \
                        ops.gemm_bf16_supported(g.stride(0), True, In_, True, Out_, In_, B):
                    # S = G^T A with the batch as the contraction index; Sb = column sums of G
                    ops.gemm_bf16(g, True, a, True, Out_, In_, B, sw.view(Out_, In_), colsum=sb, name=probe)
                else:
                    g = g.contiguous()
                    ops._probe(probe, "begin")
                    check(lib.vqaf_clipped_sum_linear(ptr(a), ptr(g), None if prescaled else ptr(coef), B,
                                                      In_, Out_, dtype_code(g),
                                                      acc if g.dtype == torch.bfloat16 else 0, ptr(sw), ptr(sb), st),
                          "vqaf_clipped_sum_linear")
                    ops._probe(probe, "end")
            else:
                check(lib.vqaf_clipped_sum_affine(ptr(a), ptr(g), None if prescaled else ptr(coef), B, g.shape[1],
                                                  dtype_code(g), acc, ptr(self._slice(S, m.weight)),
                                                  ptr(self._slice(S, m.bias)), st),
                      "vqaf_clipped_sum_affine")
        for t in tapes:  # the recorded backprops were scaled in place: they are spent
            t.g = None
        return S

    def _single_group(self):
        """(group, lo) when every private parameter sits in ONE group of the wrapped optimizer, else None."""
        groups = [g for g in self.original_optimizer.param_groups if any(id(p) in self._offsets for p in g["params"])]
        if len(self.original_optimizer.param_groups) != 1 or len(groups) != 1:
            return None
        return groups[0], min(self._offsets[id(p)] for p in groups[0]["params"] if id(p) in self._offsets)

    def _sums_with_update(self, cand, B: int, probe: str) -> bool:
        """The Linear layers' clipped sums and their noise + AdamW update in one launch; False (nothing
        launched) when the shapes are not covered."""
        import ctypes
        sg = self._single_group()
        if sg is None or any(id(t.module.weight) not in self._offsets or
                             (t.module.bias is not None and id(t.module.bias) not in self._offsets) or
                             t.a.shape[1] % 8 != 0 for t in cand):
            return False
        group, lo = sg
        n = len(cand)
        vp, lg, it = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
        b1, b2 = group["betas"]
        ops._probe(probe, "begin")
        status = load().vqaf_clipped_sum_adamw_group(
            n, vp(*[t.g.data_ptr() for t in cand]), lg(*[t.g.stride(0) for t in cand]),
            vp(*[t.a.data_ptr() for t in cand]), lg(*[t.a.stride(0) for t in cand]),
            it(*[t.g.shape[1] for t in cand]), it(*[t.a.shape[1] for t in cand]), B,
            lg(*[self._offsets[id(t.module.weight)] for t in cand]),
            lg(*[self._offsets[id(t.module.bias)] if t.module.bias is not None else -1 for t in cand]), lo,
            ptr(self._step), self.noise_multiplier * self.max_grad_norm, 1.0 / self.expected_batch_size,
            ptr(self._P), ptr(self._M), ptr(self._V), ptr(self._shadow), ptr(self._grad), float(group["lr"]), float(b1),
            float(b2), float(group["eps"]), float(group["weight_decay"]), ptr(self._step), stream_ptr(self._P.device))
        ops._probe(probe, "end")
        if status == -1:  # VQAF_E_SHAPE: too few tiles for one launch
            return False
        check(status, "vqaf_clipped_sum_adamw_group")
        return True

    @torch.no_grad()
    def _noise_scale_local(self, noise: Optional[torch.Tensor] = None):
        """grad = (S + (sigma C / sqrt(world)) z) / expected_batch_size on this rank (no exchange)."""
        S, grad = self._S, self._grad
        world, _ = self._world()
        std = self.noise_multiplier * self.max_grad_norm / math.sqrt(world)
        check(load().vqaf_noise_scale(ptr(S), ptr(noise), ptr(self._rng.state), std,
                                      1.0 / self.expected_batch_size, self._numel, ptr(grad),
                                      stream_ptr(S.device)), "vqaf_noise_scale")
        if noise is None and std != 0.0:
            self._rng.advance()
        return grad

    @torch.no_grad()
    def add_noise_and_scale(self, noise: Optional[torch.Tensor] = None):
        """grad = (S + sigma C z) / expected_batch_size.  In a process group every rank adds its
        own N(0, (sigma C)^2 / world) share BEFORE the all-reduce, so the sum carries exactly
        one N(0, (sigma C)^2).  `noise` injects z (tests)."""
        grad = self._noise_scale_local(noise)
        self.reduce_step()
        return grad

    # -- the step in three phases, so that a caller can capture the two compute phases into HIP
    #    graphs and leave the collective between them as an ordinary RCCL call ------------------
    @torch.no_grad()
    def local_step(self, noise: Optional[torch.Tensor] = None):
        """Per-rank work before the exchange: ghost norms, clip, clipped sums and -- unless the
        fused single-GPU kernel will add it -- this rank's noise share and the scaling."""
        world, _ = self._world()
        self._use_fused = self._fusable() or self._P is not None
        fuse_update = False
        if self._use_fused and world == 1 and noise is None and self.fuse_update_into_sums and \
                self._single_group() is not None and self.noise_multiplier * self.max_grad_norm >= 0.0:
            tapes = [t for t in self.module.tapes if t.g is not None]
            if tapes:
                self._buffers(tapes[0].g.device)
                self._fused_buffers(tapes[0].g.device)
                fuse_update = True
        if self._use_fused:
            tapes = [t for t in self.module.tapes if t.g is not None]
            if tapes:
                self._buffers(tapes[0].g.device)
                self._fused_buffers(tapes[0].g.device)  # the update count lives there
        self.clip_and_accumulate(fuse_update=fuse_update, count_step=self._use_fused)
        if not (self._use_fused and world == 1):
            self._noise_scale_local(noise)

    @torch.no_grad()
    def reduce_step(self):
        """The one exchange of the step: sum all-reduce of the flat gradient bucket."""
        import torch.distributed as dist
        world, _ = self._world()
        if world > 1:
            # (ranks that time-share one GPU -- the one-GPU rehearsal -- drain their launch stream first)
            from .dist import drain_if_device_is_shared
            drain_if_device_is_shared(self._grad.device, self.process_group)
            dist.all_reduce(self._grad, op=dist.ReduceOp.SUM, group=self.process_group)

    @torch.no_grad()
    def apply_step(self, noise: Optional[torch.Tensor] = None):
        """Parameter update from the (reduced) gradient."""
        result = None
        if self._use_fused:
            self._fused_apply(noise)
        else:
            result = self.original_optimizer.step()
        if self.accountant_hook is not None:
            self.accountant_hook(self)
        for t in self.module.tapes:  # a second step() without a new backward must fail loudly
            t.g = None
        # dropout call sites are numbered within a step (the number salts the mask seed, a host value): a
        # step replayed from a HIP graph then draws from the same streams as the eagerly launched one
        self.module._drop_calls = 0
        return result

    @torch.no_grad()
    def empty_step(self, device: Optional[torch.device] = None):
        """The step for an EMPTY Poisson batch: clipped sums of zero, so the update is noise only.
        Opacus takes (and accounts for) this step too; skipping it would make the number of
        accounted steps depend on the sampled data.  In a process group a rank whose shard came
        up empty still joins the all-reduce through this call."""
        self._buffers(self._S.device if self._S is not None else (device or self.params[0].device))
        self._S.zero_()
        world, _ = self._world()
        self._use_fused = self._fusable() or self._P is not None
        if not (self._use_fused and world == 1):
            self._noise_scale_local(None)
        self.reduce_step()
        return self.apply_step()

    def pre_step(self, noise: Optional[torch.Tensor] = None):
        self.clip_and_accumulate()
        self.add_noise_and_scale(noise)

    def step(self, closure=None):
        self.local_step()
        self.reduce_step()
        return self.apply_step()

    def flat_noise_layout(self):
        """(name-less) list of (offset, numel, shape) per parameter in the flat buffers."""
        return [(self._offsets[id(p)], p.numel(), tuple(p.shape)) for p in self.params]


# --------------------------------------------------------------------------------------------
# Poisson-sampled loader and the engine
# --------------------------------------------------------------------------------------------
class PoissonBatchSampler(torch.utils.data.Sampler):
    """Each step includes every sample independently with probability sample_rate
    (Opacus UniformWithReplacementSampler); batches may be empty."""

    def __init__(self, num_samples: int, sample_rate: float, steps: Optional[int] = None,
                 generator: Optional[torch.Generator] = None):
        self.num_samples, self.sample_rate = int(num_samples), float(sample_rate)
        self.steps = int(steps) if steps is not None else int(round(1.0 / sample_rate))
        self.generator = generator

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            mask = torch.rand(self.num_samples, generator=self.generator) < self.sample_rate
            yield mask.nonzero(as_tuple=False).reshape(-1).tolist()


class DistributedPoissonBatchSampler(torch.utils.data.Sampler):
    """Poisson sampling in a process group (Opacus DistributedUniformWithReplacementSampler): the
    dataset is permuted with a seed every rank shares, rank r owns the strided shard
    perm[r::world] -- shards are DISJOINT, so a sample can enter at most one rank's batch and its
    total contribution to the all-reduced sum stays <= C -- and each step includes every owned
    sample independently with probability sample_rate from a rank-private generator.  Over all
    ranks a step is exactly one Poisson sample of the whole dataset at rate sample_rate, which is
    what the accountant is stepped with."""

    def __init__(self, num_samples: int, sample_rate: float, rank: int, world: int,
                 steps: Optional[int] = None, shuffle_seed: int = 0, generator: Optional[torch.Generator] = None):
        self.total, self.sample_rate = int(num_samples), float(sample_rate)
        self.rank, self.world = int(rank), int(world)
        self.steps = int(steps) if steps is not None else int(round(1.0 / sample_rate))
        self.shuffle_seed, self.epoch = int(shuffle_seed), 0
        if generator is None:
            generator = torch.Generator().manual_seed((torch.initial_seed() + 7919 * (self.rank + 1)) % (1 << 63))
        self.generator = generator

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def owned(self) -> torch.Tensor:
        g = torch.Generator().manual_seed(self.shuffle_seed + self.epoch)
        return torch.randperm(self.total, generator=g)[self.rank::self.world]

    def __len__(self):
        return self.steps

    def __iter__(self):
        mine = self.owned()
        for _ in range(self.steps):
            mask = torch.rand(mine.numel(), generator=self.generator) < self.sample_rate
            yield mine[mask].tolist()


class DPDataLoader(torch.utils.data.DataLoader):
    """DataLoader over the same dataset with Poisson batches (make_private(poisson_sampling=True))."""

    @classmethod
    def from_data_loader(cls, dl: torch.utils.data.DataLoader, generator=None, rank: int = 0, world: int = 1,
                         shuffle_seed: int = 0):
        sample_rate = 1.0 / len(dl)
        if world > 1:
            sampler = DistributedPoissonBatchSampler(len(dl.dataset), sample_rate, rank, world, steps=len(dl),
                                                     shuffle_seed=shuffle_seed, generator=generator)
        else:
            sampler = PoissonBatchSampler(len(dl.dataset), sample_rate, steps=len(dl), generator=generator)
        collate = dl.collate_fn

        def collate_skip_empty(batch):
            return collate(batch) if len(batch) > 0 else None

        out = cls(dl.dataset, batch_sampler=sampler, num_workers=dl.num_workers,
                  collate_fn=collate_skip_empty, pin_memory=dl.pin_memory)
        out.sample_rate = sample_rate
        return out


class PrivacyEngine:
    """make_private / make_private_with_epsilon / get_epsilon as used at train.py:127-153, 209."""

    def __init__(self, *, accountant: str = "prv", secure_mode: bool = False):
        # "prv" is the default of the engine the reference constructs (Opacus 1.4.0, train.py:127)
        if accountant == "prv":
            from .prv import PRVAccountant
            self.accountant = PRVAccountant()
        elif accountant == "rdp":
            self.accountant = RDPAccountant()
        else:
            raise ValueError(f"unknown accountant {accountant!r}: 'prv' or 'rdp'")
        if secure_mode:
            raise ValueError("secure_mode (cryptographic RNG) is not implemented; the reference "
                             "passes secure_mode=False (train.py:127)")
        self.secure_mode = secure_mode

    def make_private(self, *, module: nn.Module, optimizer: torch.optim.Optimizer, data_loader,
                     noise_multiplier: float, max_grad_norm: float, batch_first: bool = True,
                     loss_reduction: str = "mean", poisson_sampling: bool = True,
                     noise_generator=None, seed: Optional[int] = None, process_group=None):
        if isinstance(module, GradSampleModule):
            gsm = module
        else:
            gsm = GradSampleModule(module, batch_first=batch_first, loss_reduction=loss_reduction)
        if data_loader is None:
            raise ValueError("data_loader is required to derive sample_rate and expected_batch_size")
        world, rank = 1, 0
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        sample_rate = 1.0 / len(data_loader)
        # the LOGICAL expected batch: one Poisson sample of the whole dataset per step, drawn
        # jointly by all ranks from disjoint shards
        expected = int(len(data_loader.dataset) * sample_rate)
        if poisson_sampling:
            loader = DPDataLoader.from_data_loader(data_loader, rank=rank, world=world,
                                                   shuffle_seed=int(seed or 0))
        elif world > 1:
            # without Poisson sampling the caller's loader decides what each rank sees; identical
            # loaders on every rank would count each sample world times against one noise draw
            raise ValueError("with several ranks make_private needs poisson_sampling=True (rank-disjoint "
                             "Poisson shards); a caller-sharded fixed loader is not accounted for here")
        else:
            loader = data_loader
        dp_opt = DPOptimizer(optimizer, gsm, noise_multiplier=noise_multiplier,
                             max_grad_norm=max_grad_norm, expected_batch_size=expected,
                             seed=seed, process_group=process_group)
        self.sample_rate = sample_rate
        dp_opt.attach_step_hook(
            lambda _o: self.accountant.step(noise_multiplier=noise_multiplier, sample_rate=sample_rate))
        return gsm, dp_opt, loader

    def make_private_with_epsilon(self, *, module, optimizer, data_loader, target_epsilon: float,
                                  target_delta: float, epochs: int, max_grad_norm: float,
                                  batch_first: bool = True, loss_reduction: str = "mean",
                                  poisson_sampling: bool = True, **kwargs):
        sample_rate = 1.0 / len(data_loader)
        sigma = get_noise_multiplier(target_epsilon=target_epsilon, target_delta=target_delta,
                                     sample_rate=sample_rate, steps=int(epochs * len(data_loader)),
                                     accountant=self.accountant.mechanism())
        return self.make_private(module=module, optimizer=optimizer, data_loader=data_loader,
                                 noise_multiplier=sigma, max_grad_norm=max_grad_norm,
                                 batch_first=batch_first, loss_reduction=loss_reduction,
                                 poisson_sampling=poisson_sampling, **kwargs)

    def get_epsilon(self, delta: float) -> float:
        return self.accountant.get_epsilon(delta)
