"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of the two encoder stacks the reference obtains from Hugging Face
(`BertModel.from_pretrained`, model/text_encoder.py:21,37-41; `SwinModel.from_pretrained` /
the ViT-B/16 shape the benchmark configs name, model/vision_encoder.py:97,111-117).  The algorithm lives in
the third-party package `transformers` (requirements.txt:76 pins 4.52.4), absent from /root/reference;
it is restated here from that package's published model definitions (`modeling_vit.py`: ViTEmbeddings,
ViTLayer, ViTPooler; `modeling_bert.py`: BertEmbeddings, BertLayer, BertPooler) as plain functions over a
state dict with the checkpoint key names of that version.

Pinned: tests/test_encoder_oracle.py runs the `transformers` package installed in this image on seeded
random weights of small configs and requires these functions to reproduce its `last_hidden_state` and
`pooler_output` (f32, rtol 1e-5); the ViT key names of the installed (newer) release are mapped by the test.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _mha(x, sd, pre, heads, key_bias=None):
    """softmax(q k^T / sqrt(dh) + bias) v with separate query/key/value Linear layers."""
    B, N, E = x.shape
    dh = E // heads
    q = F.linear(x, sd[pre + "query.weight"], sd[pre + "query.bias"]).view(B, N, heads, dh).transpose(1, 2)
    k = F.linear(x, sd[pre + "key.weight"], sd[pre + "key.bias"]).view(B, N, heads, dh).transpose(1, 2)
    v = F.linear(x, sd[pre + "value.weight"], sd[pre + "value.bias"]).view(B, N, heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    return (p @ v).transpose(1, 2).reshape(B, N, E)


def vit_forward(sd, images, heads, patch, eps=1e-12):
    """ViTModel.forward with add_pooling_layer=True: (last_hidden_state, pooler_output)."""
    x = F.conv2d(images, sd["embeddings.patch_embeddings.projection.weight"],
                 sd["embeddings.patch_embeddings.projection.bias"], stride=patch).flatten(2).transpose(1, 2)
    B = x.shape[0]
    x = torch.cat([sd["embeddings.cls_token"].expand(B, -1, -1), x], 1) + sd["embeddings.position_embeddings"]
    i = 0
    while f"encoder.layer.{i}.layernorm_before.weight" in sd:
        p = f"encoder.layer.{i}."
        E = x.shape[-1]
        y = F.layer_norm(x, (E,), sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"], eps)
        a = _mha(y, sd, p + "attention.attention.", heads)
        x = F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"]) + x
        y = F.layer_norm(x, (E,), sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"], eps)
        h = F.gelu(F.linear(y, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        x = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]) + x
        i += 1
    x = F.layer_norm(x, (x.shape[-1],), sd["layernorm.weight"], sd["layernorm.bias"], eps)
    pooled = torch.tanh(F.linear(x[:, 0], sd["pooler.dense.weight"], sd["pooler.dense.bias"]))
    return x, pooled


def bert_forward(sd, input_ids, attention_mask, heads, eps=1e-12, roberta_pad=None):
    """BertModel.forward in eval mode, token_type_ids = 0, absolute positions.  roberta_pad = the padding
    index selects RobertaModel's embeddings (`modeling_roberta.py`: create_position_ids_from_input_ids)."""
    L = input_ids.shape[1]
    if roberta_pad is None:
        pos = sd["embeddings.position_embeddings.weight"][:L]
    else:
        live = (input_ids != roberta_pad).long()
        pos = sd["embeddings.position_embeddings.weight"][torch.cumsum(live, 1) * live + roberta_pad]
    x = (sd["embeddings.word_embeddings.weight"][input_ids] + pos
         + sd["embeddings.token_type_embeddings.weight"][0])
    E = x.shape[-1]
    x = F.layer_norm(x, (E,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)
    bias = None
    if attention_mask is not None:
        bias = (1.0 - attention_mask.to(x.dtype)) * torch.finfo(x.dtype).min
    i = 0
    while f"encoder.layer.{i}.attention.self.query.weight" in sd:
        p = f"encoder.layer.{i}."
        a = _mha(x, sd, p + "attention.self.", heads, bias)
        a = F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        x = F.layer_norm(a + x, (E,), sd[p + "attention.output.LayerNorm.weight"],
                         sd[p + "attention.output.LayerNorm.bias"], eps)
        h = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        o = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        x = F.layer_norm(o + x, (E,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
        i += 1
    pooled = torch.tanh(F.linear(x[:, 0], sd["pooler.dense.weight"], sd["pooler.dense.bias"]))
    return x, pooled


def attention_reference(q, k, v, key_bias, scale):
    """f64 attention of (B,H,N,dh) operands, for the kernel-level checks."""
    s = (q.double() @ k.double().transpose(-1, -2)) * scale
    if key_bias is not None:
        s = s + key_bias.double().clamp_min(-1e30)[:, None, None, :]
    return torch.softmax(s, dim=-1) @ v.double()
