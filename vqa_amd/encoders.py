"""Vision and text encoders behind the reference's encoder contract.

The reference builds its encoders from Hugging Face models and projects them to the fusion width:
`SwinTransformer` (model/vision_encoder.py:83-127) returns `(v_seq (B,N,512), v_global (B,512))`,
`Bert` (model/text_encoder.py:14-48) returns `(q_seq (B,L,512), q_global (B,512))`; both expose
`output_dim` and are passed to `VQAModel` as classes called with `out_features=` / `hidden_dim=`.

Here the transformer stacks themselves are built on the HIP kernels of csrc/encoder.hip (attention,
LayerNorm with the residual add, GELU), their Linear layers on the library's own bf16 GEMM kernels (csrc/linear_nt.hip,
csrc/wgrad256.hip, csrc/gemm_bf16.hip; shapes those do not cover fall to torch's BLAS library).  `ViTModel` is a
ViT-B/16-shaped stack (pre-LayerNorm blocks, CLS token, tanh pooler) and `BertModel` a BERT-base-shaped
one (post-LayerNorm blocks, additive padding mask, tanh pooler).  Parameter names and shapes are those of
the Hugging Face checkpoints of the two architectures (transformers 4.52, the version the reference pins:
`encoder.layer.N.attention.attention.query.weight` for ViT, `encoder.layer.N.attention.self.query.weight`
for BERT), so a checkpoint directory on the local disk loads without renaming (`load_checkpoint`).

Activations are bf16 (the attention kernel is bf16 on MFMA with f32 accumulation); parameters may be f32
(cast per use) or bf16.  There is no CPU path: the kernels raise on CPU tensors.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._lib import check, dtype_code, load, ptr, stream_ptr


# --------------------------------------------------------------------------------------------------
# autograd wrappers of the kernels
# --------------------------------------------------------------------------------------------------
class _SelfAttention(torch.autograd.Function):
    """ctx (B,N,E) <- packed qkv (B,N,3E) bf16: dropout(softmax(q k^T / sqrt(dh) + key_bias)) v per head."""

    @staticmethod
    def forward(ctx, qkv, key_bias, heads, dropout_p, rng):
        B, N, E3 = qkv.shape
        E = E3 // 3
        if qkv.dtype != torch.bfloat16:
            raise TypeError("the attention kernel computes in bf16; cast the encoder's activations to bfloat16")
        if E % heads != 0 or E // heads != 64:
            raise ValueError(f"the attention kernel supports head size 64, got {E}/{heads}")
        qkv = qkv.contiguous()
        kb = None if key_bias is None else key_bias.to(torch.float32).contiguous()
        out = torch.empty(B, N, E, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, heads, N, dtype=torch.float32, device=qkv.device)
        drop = None
        if dropout_p > 0.0:
            drop = rng.state.clone()  # the backward regenerates the mask from the same {seed, counter}
        base = ptr(qkv)
        check(load().vqaf_attn_fwd(base, base + 2 * E, base + 4 * E, E3, ptr(kb), B, heads, N, 64, 1.0 / 8.0,
                                   float(dropout_p), ptr(drop), ptr(out), E, ptr(lse), stream_ptr(qkv.device)),
              "vqaf_attn_fwd")
        if drop is not None:
            rng.advance(1)
        empty = lse.new_empty(0)
        ctx.save_for_backward(qkv, lse, kb if kb is not None else empty, drop if drop is not None else empty)
        ctx.heads = heads
        ctx.has_bias = kb is not None
        ctx.dropout_p = float(dropout_p)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        qkv, lse, kb, drop = ctx.saved_tensors
        B, N, E3 = qkv.shape
        E = E3 // 3
        g = grad_out.contiguous()
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        dvec = torch.empty_like(lse)
        base, dbase = ptr(qkv), ptr(dqkv)
        check(load().vqaf_attn_bwd(base, base + 2 * E, base + 4 * E, E3, ptr(kb) if ctx.has_bias else None,
                                   ptr(g), E, ptr(lse), B, ctx.heads, N, 64, 1.0 / 8.0, ctx.dropout_p,
                                   ptr(drop) if ctx.dropout_p > 0.0 else None,
                                   dbase, dbase + 2 * E, dbase + 4 * E, E3, ptr(dvec), stream_ptr(qkv.device)),
              "vqaf_attn_bwd")
        return dqkv, None, None, None, None


def self_attention(qkv: torch.Tensor, key_bias: torch.Tensor | None, heads: int, dropout_p: float = 0.0,
                   rng=None) -> torch.Tensor:
    """`rng`: an ops.PhiloxState on qkv's device (required when dropout_p > 0); its counter is advanced by one."""
    if dropout_p > 0.0 and rng is None:
        raise ValueError("attention dropout needs a generator state")
    return _SelfAttention.apply(qkv, key_bias, heads, float(dropout_p), rng)


class _AddLayerNorm(torch.autograd.Function):
    """(y, s) with s = x + residual (or x), y = LayerNorm(s) * weight + bias over the last dimension."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps):
        D = x.shape[-1]
        xc = x.contiguous()
        rc = None if residual is None else residual.to(x.dtype).contiguous()
        rows = xc.numel() // D
        g = weight.detach().contiguous()
        b = bias.detach().to(g.dtype).contiguous()
        y = torch.empty_like(xc)
        s = torch.empty_like(xc) if rc is not None else xc
        stats = torch.empty(2, rows, dtype=torch.float32, device=x.device)
        check(load().vqaf_layernorm_fwd(ptr(xc), ptr(rc), ptr(g), ptr(b), dtype_code(g), rows, D, float(eps),
                                        dtype_code(xc), ptr(s) if rc is not None else None, ptr(y), ptr(stats[0]),
                                        ptr(stats[1]), stream_ptr(x.device)), "vqaf_layernorm_fwd")
        ctx.save_for_backward(s, g, stats)
        ctx.has_res = rc is not None
        ctx.set_materialize_grads(False)
        return (y, s) if rc is not None else y

    @staticmethod
    def backward(ctx, gy, gs=None):
        s, g, stats = ctx.saved_tensors
        D = s.shape[-1]
        rows = s.numel() // D
        if gy is None:  # only the sum was used downstream
            return gs, (gs if ctx.has_res else None), None, None, None
        gy = gy.to(s.dtype).contiguous()
        gsc = None if gs is None else gs.to(s.dtype).contiguous()
        dx = torch.empty_like(s)
        dgb = torch.empty(2, D, dtype=g.dtype, device=s.device)
        nbytes = load().vqaf_layernorm_bwd_workspace_bytes(rows, D)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=s.device)
        check(load().vqaf_layernorm_bwd(ptr(gy), ptr(s), ptr(g), dtype_code(g), ptr(stats[0]), ptr(stats[1]), ptr(gsc),
                                        rows, D, dtype_code(s), ptr(dx), ptr(dgb[0]), ptr(dgb[1]), ptr(ws), nbytes,
                                        stream_ptr(s.device)), "vqaf_layernorm_bwd")
        return dx, (dx if ctx.has_res else None), dgb[0], dgb[1], None


def add_layer_norm(x, residual, weight, bias, eps):
    """Returns (LayerNorm(x + residual), x + residual); residual may be None (then the second item is x)."""
    if residual is None:
        return _AddLayerNorm.apply(x, None, weight, bias, eps), x
    return _AddLayerNorm.apply(x, residual, weight, bias, eps)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        y = torch.empty_like(xc)
        check(load().vqaf_gelu_fwd(ptr(xc), xc.numel(), dtype_code(xc), ptr(y), stream_ptr(x.device)), "vqaf_gelu_fwd")
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    def backward(ctx, gy):
        (xc,) = ctx.saved_tensors
        gy = gy.to(xc.dtype).contiguous()
        dx = torch.empty_like(xc)
        check(load().vqaf_gelu_bwd(ptr(xc), ptr(gy), xc.numel(), dtype_code(xc), ptr(dx), stream_ptr(xc.device)),
              "vqaf_gelu_bwd")
        return dx


def gelu(x: torch.Tensor) -> torch.Tensor:
    return _Gelu.apply(x)


# token count from which the weight gradient runs on the large-tile kernel (below it the contraction is too
# short to split into enough slices for 128 x 128 tiles)
_WGRAD_TN_MIN_TOKENS = 4096


def _big_tiles(rows: int, cols: int) -> bool:
    """Enough 128 x 128 output tiles to fill the chip with the large-tile kernel (vqaf_linear_nt); smaller
    products run faster on the 128 x 64 / 64 x 64 tiles of vqaf_gemm_bf16."""
    return ((rows + 127) // 128) * ((cols + 127) // 128) >= 192


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b for bf16 activations, every product on the library's own bf16 MFMA kernels.
    With enough 128 x 128 output tiles to fill the chip the forward and the input gradient run on the
    large-tile kernel (vqaf_linear_nt; the input gradient reads W as stored, through transposing LDS reads),
    and the weight gradient, whose contraction index (the token) is the slow index of both operands, on its
    k-major twin (vqaf_wgrad_tn), which also delivers the bias gradient from the fragments it multiplies.
    Smaller products (the text encoder at 30 tokens per sample, the projections of the pooled vector) run on
    vqaf_gemm_bf16.  Shapes neither covers (pitches that are not multiples of 8) go to torch."""

    @staticmethod
    def forward(ctx, x, w, b):
        from . import ops
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.bdtype = None if b is None else b.dtype
        out_f, in_f = w.shape
        x2 = x.reshape(-1, in_f)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = x2.shape[0]
        wc = w if w.is_contiguous() else w.contiguous()
        if _big_tiles(T, out_f) and ops.linear_nt_supported(x2, wc):
            y = ops.linear_nt(x2, wc, b)
        elif ops.gemm_bf16_supported(in_f, False, in_f, False, T, out_f, in_f):
            y = torch.empty(T, out_f, dtype=torch.bfloat16, device=x.device)
            ops.gemm_bf16(x2, False, wc, False, T, out_f, in_f, y, bias=b)
        else:
            return F.linear(x, w, b)
        return y.view(*x.shape[:-1], out_f)

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x, w = ctx.saved_tensors
        out_f, in_f = w.shape
        gy2 = gy.reshape(-1, out_f)
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        x2 = x.reshape(-1, in_f)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = x2.shape[0]
        gx = None
        if ctx.needs_input_grad[0]:
            wc = w if w.is_contiguous() else w.contiguous()
            if _big_tiles(T, in_f) and ops.linear_nt_supported(gy2, wc, True):
                gx = ops.linear_nt(gy2, wc, None, True)
            elif ops.gemm_bf16_supported(out_f, False, in_f, True, T, in_f, out_f):
                gx = torch.empty(T, in_f, dtype=torch.bfloat16, device=gy.device)
                ops.gemm_bf16(gy2, False, wc, True, T, in_f, out_f, gx)
            else:
                gx = torch.mm(gy2, w)
            gx = gx.view(x.shape)
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if T >= _WGRAD_TN_MIN_TOKENS and ops.wgrad_tn_supported(gy2, x2):
                gw, gb = ops.wgrad_tn(gy2, x2, w.dtype, ctx.bdtype)
            elif ops.gemm_bf16_supported(out_f, True, in_f, True, out_f, in_f, T):
                # An f32 output lets the kernel split the contraction over workgroups when the output has
                # too few tiles to fill the chip; enough tiles (>= 192 of 128 x 64) go straight to bf16.
                direct = ((out_f + 127) // 128) * ((in_f + 63) // 64) >= 192
                gw = torch.empty(out_f, in_f, dtype=torch.bfloat16 if direct else torch.float32, device=x.device)
                cs = torch.empty(out_f, dtype=torch.float32, device=x.device) if ctx.has_bias else None
                ops.gemm_bf16(gy2, True, x2, True, out_f, in_f, T, gw, colsum=cs)
                if not direct:
                    gw = gw.to(w.dtype)
                gb = None if cs is None else cs.to(ctx.bdtype)
            else:  # row pitches the kernels do not cover (not multiples of 8): library GEMM
                gw = torch.mm(gy2.t(), x2)
                gb = gy2.sum(0).to(ctx.bdtype) if ctx.has_bias else None
        return gx, gw, gb


def _affine(x, w, b):
    if x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16:
        return _LinearFn.apply(x, w, b)
    return F.linear(x, w, b)


def _linear(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    w, b = lin.weight, lin.bias
    if w.dtype != x.dtype:
        w = w.to(x.dtype)
        b = None if b is None else b.to(x.dtype)
    return _affine(x, w, b)


def _packed_qkv(x, q: nn.Linear, k: nn.Linear, v: nn.Linear) -> torch.Tensor:
    """One GEMM for the three projections (the checkpoint keeps them as three Linear layers)."""
    w = torch.cat([q.weight, k.weight, v.weight], 0)
    b = torch.cat([q.bias, k.bias, v.bias], 0)
    if w.dtype != x.dtype:
        w, b = w.to(x.dtype), b.to(x.dtype)
    return _affine(x, w, b)


# --------------------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------------------
@dataclass
class ViTConfig:
    """google/vit-base-patch16-224 shape by default; image_size=384 gives 577 tokens."""
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    image_size: int = 224
    patch_size: int = 16
    num_channels: int = 3
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.0
    attention_probs_dropout_prob: float = 0.0
    initializer_range: float = 0.02


@dataclass
class BertConfig:
    """bert-base-uncased shape by default."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    pad_token_id: int = 0
    initializer_range: float = 0.02
    # RoBERTa numbers positions from pad_token_id + 1 and skips padding tokens (model_type "roberta")
    roberta_positions: bool = False


def roberta_config(**kw) -> "BertConfig":
    """roberta-base shape: the BERT stack with 50265 tokens, 514 positions counted from the padding index,
    one token type and LayerNorm eps 1e-5."""
    base = dict(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                pad_token_id=1, roberta_positions=True)
    base.update(kw)
    return BertConfig(**base)


def _config_from_json(cls, path):
    with open(path) as f:
        raw = json.load(f)
    fields = {k: raw[k] for k in cls.__dataclass_fields__ if k in raw}
    if cls is BertConfig and raw.get("model_type") == "roberta":
        fields["roberta_positions"] = True
    return cls(**fields)


def _init_weights(module: nn.Module, std: float) -> None:
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=std)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, std=std)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


def _dropout_rng(module: nn.Module, device: torch.device):
    """The module's generator state for attention dropout (made on first use, seeded from torch's seed,
    rank-dependent so that data-parallel replicas draw different masks)."""
    from . import ops
    st = getattr(module, "_attn_rng", None)
    if st is None or st.state.device != device:
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        st = ops.PhiloxState(device, stream_salt=41 + 1009 * rank)
        module._attn_rng = st
    return st


# --------------------------------------------------------------------------------------------------
# ViT
# --------------------------------------------------------------------------------------------------
class _Named(nn.Module):
    """A module that only groups children under a checkpoint name."""


class _ViTPatchEmbeddings(nn.Module):
    def __init__(self, cfg: ViTConfig):
        super().__init__()
        self.projection = nn.Conv2d(cfg.num_channels, cfg.hidden_size, cfg.patch_size, stride=cfg.patch_size)
        self.patch = cfg.patch_size

    def forward(self, images):
        # non-overlapping patches: the strided convolution is one GEMM over (B*gh*gw, C*P*P) patch rows
        B, C, Hh, Ww = images.shape
        P = self.patch
        gh, gw = Hh // P, Ww // P
        rows = images.reshape(B, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * P * P)
        w = self.projection.weight.reshape(self.projection.weight.shape[0], -1)
        b = self.projection.bias
        if w.dtype != rows.dtype:
            w, b = w.to(rows.dtype), b.to(rows.dtype)
        return _affine(rows, w, b)


class _ViTEmbeddings(nn.Module):
    def __init__(self, cfg: ViTConfig):
        super().__init__()
        n = (cfg.image_size // cfg.patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.hidden_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, n + 1, cfg.hidden_size))
        self.patch_embeddings = _ViTPatchEmbeddings(cfg)
        nn.init.trunc_normal_(self.cls_token, std=cfg.initializer_range)
        nn.init.trunc_normal_(self.position_embeddings, std=cfg.initializer_range)
        self.p = cfg.hidden_dropout_prob

    def forward(self, images):
        x = self.patch_embeddings(images)
        B = x.shape[0]
        if x.shape[1] + 1 != self.position_embeddings.shape[1]:
            raise ValueError(f"{x.shape[1]} patches but {self.position_embeddings.shape[1] - 1} position embeddings: "
                             "build the encoder with the image size it is fed (no position interpolation)")
        x = torch.cat([self.cls_token.to(x.dtype).expand(B, -1, -1), x], 1) + self.position_embeddings.to(x.dtype)
        return F.dropout(x, self.p, self.training)


class _ViTSelfAttention(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(E, E), nn.Linear(E, E), nn.Linear(E, E)


class _Dense(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.dense = nn.Linear(i, o)


class _ViTAttention(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.attention = _ViTSelfAttention(E)
        self.output = _Dense(E, E)


class _ViTLayer(nn.Module):
    def __init__(self, cfg: ViTConfig):
        super().__init__()
        E = cfg.hidden_size
        self.attention = _ViTAttention(E)
        self.intermediate = _Dense(E, cfg.intermediate_size)
        self.output = _Dense(cfg.intermediate_size, E)
        self.layernorm_before = nn.LayerNorm(E, eps=cfg.layer_norm_eps)
        self.layernorm_after = nn.LayerNorm(E, eps=cfg.layer_norm_eps)
        self.heads, self.eps, self.p = cfg.num_attention_heads, cfg.layer_norm_eps, cfg.hidden_dropout_prob
        self.p_attn = cfg.attention_probs_dropout_prob

    def forward(self, y, s, rng=None):
        """y = layernorm_before(s) of this layer; returns (layernorm of the NEXT consumer's input, stream)
        as a pair so that every residual add runs inside the LayerNorm kernel that follows it."""
        a = self.attention.attention
        ctx = self_attention(_packed_qkv(y, a.query, a.key, a.value), None, self.heads,
                             self.p_attn if self.training else 0.0, rng)
        att = F.dropout(_linear(ctx, self.attention.output.dense), self.p, self.training)
        y2, s2 = add_layer_norm(att, s, self.layernorm_after.weight, self.layernorm_after.bias, self.eps)
        h = gelu(_linear(y2, self.intermediate.dense))
        o = F.dropout(_linear(h, self.output.dense), self.p, self.training)
        return o, s2


class _Layers(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layer = nn.ModuleList(layers)


class _Pooler(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.dense = nn.Linear(E, E)

    def forward(self, x):
        return torch.tanh(_linear(x[:, 0], self.dense))


class ViTModel(nn.Module):
    """`last_hidden_state (B, N+1, E), pooler_output (B, E)` of a ViT with the layout of
    google/vit-base-patch16-224 (pre-LayerNorm blocks; final LayerNorm; pooler = tanh(dense(CLS)))."""

    def __init__(self, config: ViTConfig | None = None, compute_dtype=torch.bfloat16):
        super().__init__()
        self.config = cfg = config or ViTConfig()
        self.compute_dtype = compute_dtype
        self.embeddings = _ViTEmbeddings(cfg)
        self.encoder = _Layers([_ViTLayer(cfg) for _ in range(cfg.num_hidden_layers)])
        self.layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.pooler = _Pooler(cfg.hidden_size)
        _init_weights(self, cfg.initializer_range)

    def forward(self, pixel_values):
        x = self.embeddings(pixel_values.to(self.compute_dtype))
        # block i:  s <- s + attn(LN1_i(s));  s <- s + mlp(LN2_i(s)).  Each add is done by the LayerNorm kernel
        # that consumes the sum, so the loop carries (pending branch output, stream before the add).
        pending, s = None, x
        rng = _dropout_rng(self, x.device) if self.training and self.config.attention_probs_dropout_prob > 0 else None
        for lyr in self.encoder.layer:
            y, s = add_layer_norm(s if pending is None else pending, None if pending is None else s,
                                  lyr.layernorm_before.weight, lyr.layernorm_before.bias, lyr.eps)
            pending, s = lyr(y, s, rng)
        h, _ = add_layer_norm(pending, s, self.layernorm.weight, self.layernorm.bias, self.config.layer_norm_eps)
        return h, self.pooler(h)


# --------------------------------------------------------------------------------------------------
# BERT
# --------------------------------------------------------------------------------------------------
class _EmbeddingSum(torch.autograd.Function):
    """word[ids] + position[pos_ids] + token_type[type_ids] in one launch (vqaf_embed_fwd); the backward gives
    each table the sum of the gradient rows of the tokens that name a row, added in token order
    (vqaf_embed_bwd), the padding row of the word table staying zero as nn.Embedding(padding_idx=...) keeps it."""

    @staticmethod
    def forward(ctx, word_w, pos_w, type_w, ids, pos_ids, type_ids, out_dtype, pad_idx):
        E = word_w.shape[1]
        out = torch.empty(*ids.shape, E, dtype=out_dtype, device=ids.device)
        check(load().vqaf_embed_fwd(ptr(ids), ptr(word_w), word_w.shape[0], ptr(pos_ids), ptr(pos_w), pos_w.shape[0],
                                    ptr(type_ids), ptr(type_w), type_w.shape[0], ids.numel(), E, dtype_code(word_w),
                                    ptr(out), dtype_code(out), stream_ptr(ids.device)), "vqaf_embed_fwd")
        ctx.save_for_backward(ids, pos_ids, type_ids)
        ctx.rows = (word_w.shape[0], pos_w.shape[0], type_w.shape[0])
        ctx.pdtype, ctx.pad = word_w.dtype, pad_idx
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        E = g.shape[-1]
        grads = []
        for k, (idx, rows) in enumerate(zip(ctx.saved_tensors, ctx.rows)):
            if not ctx.needs_input_grad[k]:
                grads.append(None)
                continue
            tg = torch.empty(rows, E, dtype=ctx.pdtype, device=g.device)
            pad = ctx.pad if (k == 0 and ctx.pad is not None) else -1
            check(load().vqaf_embed_bwd(ptr(idx), idx.numel(), ptr(g), dtype_code(g), E, rows, pad, ptr(tg),
                                        dtype_code(tg), stream_ptr(g.device)), "vqaf_embed_bwd")
            grads.append(tg)
        return grads[0], grads[1], grads[2], None, None, None, None, None


class _BertEmbeddings(nn.Module):
    def __init__(self, cfg: BertConfig):
        super().__init__()
        E = cfg.hidden_size
        self.word_embeddings = nn.Embedding(cfg.vocab_size, E, padding_idx=cfg.pad_token_id)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, E)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, E)
        self.LayerNorm = nn.LayerNorm(E, eps=cfg.layer_norm_eps)
        self.eps, self.p = cfg.layer_norm_eps, cfg.hidden_dropout_prob
        self.pad, self.roberta = cfg.pad_token_id, cfg.roberta_positions

    def forward(self, input_ids, dtype):
        B, L = input_ids.shape
        ww, pw, tw = self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight
        fused = (input_ids.is_cuda and input_ids.numel() <= 32768 and ww.shape[1] % 8 == 0 and
                 ww.dtype == pw.dtype == tw.dtype and ww.dtype in (torch.float32, torch.bfloat16) and
                 dtype in (torch.float32, torch.bfloat16))
        if self.roberta:
            # RobertaEmbeddings: position = pad + (1-based index among the non-padding tokens), pad for padding
            live = (input_ids != self.pad).to(torch.long)
            pos_ids = torch.cumsum(live, 1) * live + self.pad
        else:
            pos_ids = None
        if fused:
            ids = input_ids.contiguous()
            if pos_ids is None:
                pos_ids = torch.arange(L, device=ids.device).repeat(B, 1)
            x = _EmbeddingSum.apply(ww, pw, tw, ids, pos_ids.contiguous(), torch.zeros_like(ids), dtype, self.pad)
        else:
            pos = (self.position_embeddings(pos_ids) if pos_ids is not None else pw[:L]) + tw[0]
            x = (self.word_embeddings(input_ids) + pos).to(dtype)
        y, _ = add_layer_norm(x, None, self.LayerNorm.weight, self.LayerNorm.bias, self.eps)
        return F.dropout(y, self.p, self.training)


class _BertSelfAttention(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(E, E), nn.Linear(E, E), nn.Linear(E, E)


class _BertSelfOutput(nn.Module):
    def __init__(self, i, o, eps):
        super().__init__()
        self.dense = nn.Linear(i, o)
        self.LayerNorm = nn.LayerNorm(o, eps=eps)


class _BertAttention(nn.Module):
    def __init__(self, E, eps):
        super().__init__()
        self.self = _BertSelfAttention(E)
        self.output = _BertSelfOutput(E, E, eps)


class _LoraPair(nn.Module):
    """One LoRA adapter of a Linear (in -> out): lora_A (r, in), lora_B (out, r), as the PEFT package names and
    initialises them (A: Kaiming-uniform with a = sqrt(5), B: zeros, so the adapted layer starts as the base one)."""

    def __init__(self, in_f: int, out_f: int, r: int):
        super().__init__()
        self.lora_A = nn.Linear(in_f, r, bias=False)
        self.lora_B = nn.Linear(r, out_f, bias=False)
        nn.init.kaiming_uniform_(self.lora_A.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B.weight)


class _QkvLora(nn.Module):
    """LoRA adapters on the query / key / value projections of one attention block (model/text_encoder.py:91-169
    configures r = 8, alpha = 16, dropout 0.1 on ["query", "value"]):
        y_t = x W_t^T + b_t + (alpha / r) * dropout(x) A_t^T B_t^T      for every adapted projection t.
    The adapters of a block share two small products on the packed (3E-wide) output: dropout(x) against the
    stacked A matrices, then against a (3E, n r) matrix that holds each scaled B in its projection's rows and
    zeros elsewhere."""
    ORDER = ("query", "key", "value")

    def __init__(self, E: int, r: int, alpha: float, dropout: float, targets):
        super().__init__()
        bad = [t for t in targets if t not in self.ORDER]
        if bad:
            raise NotImplementedError(f"LoRA target modules {bad}: only query / key / value are adapted here")
        self.targets = [t for t in self.ORDER if t in targets]
        self.adapters = nn.ModuleDict({t: _LoraPair(E, E, r) for t in self.targets})
        self.E, self.r, self.scaling, self.p = E, r, float(alpha) / r, float(dropout)

    def forward(self, x, training: bool):
        E, r = self.E, self.r
        xd = F.dropout(x, self.p, training) if self.p > 0 else x
        a_cat = torch.cat([self.adapters[t].lora_A.weight for t in self.targets], 0).to(x.dtype)   # (n r, E)
        b_cat = x.new_zeros(3 * E, len(self.targets) * r)
        for j, t in enumerate(self.targets):
            o = self.ORDER.index(t) * E
            b_cat[o:o + E, j * r:(j + 1) * r] = self.adapters[t].lora_B.weight.to(x.dtype) * self.scaling
        return _affine(_affine(xd, a_cat, None), b_cat, None)


class _BertLayer(nn.Module):
    def __init__(self, cfg: BertConfig):
        super().__init__()
        E = cfg.hidden_size
        self.attention = _BertAttention(E, cfg.layer_norm_eps)
        self.intermediate = _Dense(E, cfg.intermediate_size)
        self.output = _BertSelfOutput(cfg.intermediate_size, E, cfg.layer_norm_eps)
        self.heads, self.eps, self.p = cfg.num_attention_heads, cfg.layer_norm_eps, cfg.hidden_dropout_prob
        self.p_attn = cfg.attention_probs_dropout_prob
        self.lora = None  # a _QkvLora once add_lora() has adapted the block

    def forward(self, x, key_bias, rng=None):
        a = self.attention.self
        qkv = _packed_qkv(x, a.query, a.key, a.value)
        if self.lora is not None:
            qkv = qkv + self.lora(x, self.training)
        ctx = self_attention(qkv, key_bias, self.heads,
                             self.p_attn if self.training else 0.0, rng)
        att = F.dropout(_linear(ctx, self.attention.output.dense), self.p, self.training)
        ln = self.attention.output.LayerNorm
        x1, _ = add_layer_norm(att, x, ln.weight, ln.bias, self.eps)
        h = gelu(_linear(x1, self.intermediate.dense))
        o = F.dropout(_linear(h, self.output.dense), self.p, self.training)
        ln = self.output.LayerNorm
        x2, _ = add_layer_norm(o, x1, ln.weight, ln.bias, self.eps)
        return x2


class BertModel(nn.Module):
    """`last_hidden_state (B, L, E), pooler_output (B, E)` of a BERT encoder with the layout of
    bert-base-uncased (post-LayerNorm blocks; absolute positions; token type 0)."""

    def __init__(self, config: BertConfig | None = None, compute_dtype=torch.bfloat16):
        super().__init__()
        self.config = cfg = config or BertConfig()
        self.compute_dtype = compute_dtype
        self.embeddings = _BertEmbeddings(cfg)
        self.encoder = _Layers([_BertLayer(cfg) for _ in range(cfg.num_hidden_layers)])
        self.pooler = _Pooler(cfg.hidden_size)
        _init_weights(self, cfg.initializer_range)

    def forward(self, input_ids, attention_mask=None):
        x = self.embeddings(input_ids, self.compute_dtype)
        key_bias = None
        if attention_mask is not None:
            # the extended mask of BertModel: 0 where attended, a large negative number on padding
            key_bias = (1.0 - attention_mask.to(torch.float32)) * -1.0e30
        rng = _dropout_rng(self, x.device) if self.training and self.config.attention_probs_dropout_prob > 0 else None
        for lyr in self.encoder.layer:
            x = lyr(x, key_bias, rng)
        return x, self.pooler(x)


# --------------------------------------------------------------------------------------------------
# checkpoints on the local disk
# --------------------------------------------------------------------------------------------------
def _read_state(path: str) -> dict:
    """Tensors of a checkpoint directory or file: model.safetensors, or pytorch_model.bin read with
    weights_only=True (nothing in the file is executed)."""
    if os.path.isdir(path):
        for name in ("model.safetensors", "pytorch_model.bin"):
            if os.path.exists(os.path.join(path, name)):
                path = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def load_checkpoint(model: nn.Module, path: str, prefixes=("vit.", "bert.", "roberta.")) -> nn.Module:
    """Loads a locally stored Hugging Face checkpoint of the same architecture into `model`.
    Task-head prefixes (`vit.`, `bert.`, `roberta.`) are stripped; LayerNorm `gamma`/`beta` (old TF-converted BERT
    files) are renamed; tensors the model does not have (`cls.*`, `classifier.*`, `position_ids`) are
    ignored; a tensor the model has but the file lacks is an error."""
    raw = _read_state(path)
    own = model.state_dict()
    state = {}
    for k, v in raw.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
                break
        k = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
        if k in own:
            state[k] = v
    missing = [k for k in own if k not in state]
    if missing:
        raise KeyError(f"checkpoint {path} lacks {len(missing)} tensors of the model, e.g. {missing[:3]}")
    model.load_state_dict(state)
    return model


# --------------------------------------------------------------------------------------------------
# the encoder classes VQAModel takes
# --------------------------------------------------------------------------------------------------
class ViT(nn.Module):
    """Vision encoder with the reference's contract (model/vision_encoder.py:83-127):
    `forward(images (B,3,S,S)) -> v_seq (B, N, out_features), v_global (B, out_features)` where the
    sequence is every token of the last hidden state and the global feature is the pooler output."""

    def __init__(self, model_name: str | None = None, out_features: int = 512, config: ViTConfig | None = None,
                 compute_dtype=torch.bfloat16):
        super().__init__()
        if model_name is not None and config is None and os.path.exists(os.path.join(model_name, "config.json")):
            config = _config_from_json(ViTConfig, os.path.join(model_name, "config.json"))
        self.vit = ViTModel(config, compute_dtype)
        if model_name is not None:
            load_checkpoint(self.vit, model_name)
        E = self.vit.config.hidden_size
        self.seq_projection = nn.Linear(E, out_features)
        self.global_projection = nn.Linear(E, out_features)
        self.output_dim = out_features

    def forward(self, images):
        h, pooled = self.vit(images)
        dt = self.seq_projection.weight.dtype  # the stack computes in bf16; results leave in the parameters' dtype
        return _linear(h, self.seq_projection).to(dt), _linear(pooled, self.global_projection).to(dt)


class Bert(nn.Module):
    """Text encoder with the reference's contract (model/text_encoder.py:14-48):
    `forward(input_ids, attention_mask) -> q_seq (B, L, hidden_dim), q_global (B, hidden_dim)`.
    `model_name` is a LOCAL checkpoint directory (there is no hub access); None = random init."""

    def __init__(self, model_name: str | None = None, hidden_dim: int = 512, config: BertConfig | None = None,
                 compute_dtype=torch.bfloat16):
        super().__init__()
        if model_name is not None and config is None and os.path.exists(os.path.join(model_name, "config.json")):
            config = _config_from_json(BertConfig, os.path.join(model_name, "config.json"))
        self.bert = BertModel(config, compute_dtype)
        if model_name is not None:
            load_checkpoint(self.bert, model_name)
        E = self.bert.config.hidden_size
        self.seq_projection = nn.Linear(E, hidden_dim)
        self.global_projection = nn.Linear(E, hidden_dim)
        self.output_dim = hidden_dim

    def forward(self, input_ids, attention_mask):
        h, pooled = self.bert(input_ids, attention_mask)
        dt = self.seq_projection.weight.dtype
        return _linear(h, self.seq_projection).to(dt), _linear(pooled, self.global_projection).to(dt)


class RoBerta(Bert):
    """The reference's `RoBerta` wrapper (model/text_encoder.py:51-89): the same contract and the same
    attribute names (`self.bert` holds the stack), on a roberta-base-shaped encoder."""

    def __init__(self, model_name: str | None = None, hidden_dim: int = 512, config: BertConfig | None = None,
                 compute_dtype=torch.bfloat16):
        if config is None and not (model_name is not None and os.path.exists(os.path.join(model_name, "config.json"))):
            config = roberta_config()
        super().__init__(model_name, hidden_dim, config, compute_dtype)


def add_lora(stack: "BertModel", r: int = 8, alpha: float = 16, dropout: float = 0.1, target_modules=None) -> None:
    """Freezes every parameter of `stack` and adapts the attention projections named in `target_modules`
    (default ["query", "value"]) of every block with trainable LoRA pairs: what
    `get_peft_model(base, LoraConfig(r, lora_alpha, target_modules, lora_dropout, bias="none"))` sets up."""
    targets = list(target_modules or ["query", "value"])
    for p in stack.parameters():
        p.requires_grad_(False)
    for lyr in stack.encoder.layer:
        lyr.lora = _QkvLora(stack.config.hidden_size, r, alpha, dropout, targets)


def merged_lora_weights(stack: "BertModel") -> dict:
    """{parameter name: W + (alpha / r) B A} for every adapted projection: the plain weights an adapter-free stack
    needs to compute the same function (dropout off)."""
    out = {}
    for i, lyr in enumerate(stack.encoder.layer):
        if lyr.lora is None:
            continue
        for t in lyr.lora.targets:
            pair, lin = lyr.lora.adapters[t], getattr(lyr.attention.self, t)
            out[f"encoder.layer.{i}.attention.self.{t}.weight"] = (
                lin.weight.detach().float() + lyr.lora.scaling * pair.lora_B.weight.detach().float()
                @ pair.lora_A.weight.detach().float()).to(lin.weight.dtype)
    return out


class BertQLoRA(Bert):
    """The reference's `BertQLoRA` wrapper (model/text_encoder.py:91-129): the `Bert` contract on a frozen stack
    with LoRA adapters (r = 8, alpha = 16, dropout 0.1 on query and value by default); only the adapters and the
    two projections train.  The adapters are this module's own (the `peft` package is not involved)."""

    def __init__(self, model_name: str | None = None, hidden_dim: int = 512, lora_r: int = 8, lora_alpha: float = 16,
                 lora_dropout: float = 0.1, target_modules=None, config: BertConfig | None = None,
                 compute_dtype=torch.bfloat16):
        super().__init__(model_name, hidden_dim, config, compute_dtype)
        add_lora(self.bert, lora_r, lora_alpha, lora_dropout, target_modules)


class RoBertaQLoRA(RoBerta):
    """The reference's `RoBertaQLoRA` wrapper (model/text_encoder.py:132-169), as BertQLoRA on the RoBERTa stack."""

    def __init__(self, model_name: str | None = None, hidden_dim: int = 512, lora_r: int = 8, lora_alpha: float = 16,
                 lora_dropout: float = 0.1, target_modules=None, config: BertConfig | None = None,
                 compute_dtype=torch.bfloat16):
        super().__init__(model_name, hidden_dim, config, compute_dtype)
        add_lora(self.bert, lora_r, lora_alpha, lora_dropout, target_modules)
