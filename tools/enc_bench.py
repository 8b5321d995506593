"""Developer tool: forward+backward of the full model (ViT-B/16 + BERT-base + CoAttention with pruning + classifier),
non-private, bf16, batch 64.  args: [B] [image_size] [mode: ours|hf]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import encoders, model as vm  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 224
mode = sys.argv[3] if len(sys.argv) > 3 else "ours"
dev = torch.device("cuda:0")
torch.manual_seed(0)


def vit_cls(out_features=512):
    return encoders.ViT(out_features=out_features, config=encoders.ViTConfig(image_size=S))


if mode == "ours":
    m = vm.VQAModel(vit_cls, encoders.Bert, fusion_type="co_attention", num_classes=430, use_token_pruning=True,
                    prune_ratio=0.5)
else:
    os.environ["HF_HUB_OFFLINE"] = "1"
    import transformers

    class HV(torch.nn.Module):
        def __init__(self, out_features=512):
            super().__init__()
            self.vit = transformers.ViTModel(transformers.ViTConfig(image_size=S), add_pooling_layer=True)
            self.seq_projection = torch.nn.Linear(768, out_features)
            self.global_projection = torch.nn.Linear(768, out_features)
            self.output_dim = out_features

        def forward(self, x):
            o = self.vit(pixel_values=x)
            return self.seq_projection(o.last_hidden_state), self.global_projection(o.pooler_output)

    class HB(torch.nn.Module):
        def __init__(self, hidden_dim=512):
            super().__init__()
            self.bert = transformers.BertModel(transformers.BertConfig())
            self.seq_projection = torch.nn.Linear(768, hidden_dim)
            self.global_projection = torch.nn.Linear(768, hidden_dim)
            self.output_dim = hidden_dim

        def forward(self, ids, mask):
            o = self.bert(input_ids=ids, attention_mask=mask)
            return self.seq_projection(o.last_hidden_state), self.global_projection(o.pooler_output)

    m = vm.VQAModel(HV, HB, fusion_type="co_attention", num_classes=430, use_token_pruning=True, prune_ratio=0.5)
m = m.to(dev).to(torch.bfloat16).train()
imgs = torch.randn(B, 3, S, S, device=dev, dtype=torch.bfloat16)
ids = torch.randint(1, 30000, (B, 30), device=dev)
mask = torch.ones(B, 30, dtype=torch.long, device=dev)
y = torch.randint(0, 430, (B,), device=dev)


def step():
    m.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(m(imgs, ids, mask).float(), y)
    loss.backward()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{mode} B={B} S={S}: {dt * 1e3:.2f} ms/step eager = {B / dt:.0f} samples/s")
