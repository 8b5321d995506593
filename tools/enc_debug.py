"""Developer tool: kernel-level check of the attention backward on the operands of a BERT stack."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import encoders
from oracle import encoder_oracle as eo
DEV = torch.device("cuda:0")
torch.manual_seed(1)
cfg = encoders.BertConfig(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                          vocab_size=101, max_position_embeddings=40, hidden_dropout_prob=0.0)
m = encoders.BertModel(cfg)
with torch.no_grad():
    for p in m.parameters():
        p.copy_(torch.randn_like(p) * 0.08)
m.to(DEV)
ids = torch.randint(1, 101, (4, 30)); mask = torch.ones(4, 30, dtype=torch.long); mask[0, 12:] = 0; mask[2, 25:] = 0
rec = []
orig = encoders._SelfAttention.apply
def spy(qkv, kb, H):
    qkv.retain_grad()
    out = orig(qkv, kb, H)
    out.retain_grad()
    rec.append((qkv, kb, out))
    return out
encoders.self_attention = spy
h, pooled = m(ids.to(DEV), mask.to(DEV))
live = mask.bool()
cot = torch.randn(h.shape, generator=torch.Generator().manual_seed(2)).double() * live[..., None]
((h.double() * cot.to(DEV)).sum() + pooled.double().sum()).backward()
for i, (qkv, kb, out) in enumerate(rec):
    B, N, E3 = qkv.shape; H = 2
    x64 = qkv.detach().double().cpu().requires_grad_()
    q, k, v = x64.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = eo.attention_reference(q, k, v, kb.cpu(), 0.125).transpose(1, 2).reshape(B, N, H * 64)
    ref.backward(out.grad.double().cpu())
    g = qkv.grad.double().cpu().view(B, N, 3, H * 64); r = x64.grad.view(B, N, 3, H * 64)
    for j, nm in enumerate("qkv"):
        print(i, nm, "rel %.4f" % float((g[:, :, j] - r[:, :, j]).norm() / r[:, :, j].norm()), "norm %.3e" % float(r[:, :, j].norm()),
              "dout norm %.3e" % float(out.grad.double().norm()))
    print(i, "out rel", float((out.detach().double().cpu() - ref.detach()).norm() / ref.detach().norm()))
