"""Developer tool: per-step times of the whole model on the transformers package's encoders (does the convolution
library settle after its first calls?)."""
import os, sys, time
os.environ["HF_HUB_OFFLINE"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vqa_amd import model as vm, ops
dev = torch.device("cuda:0")
B = 64
imgs = torch.randn(B, 3, 224, 224, device=dev, dtype=torch.bfloat16)
ids = torch.randint(1, 30000, (B, 30), device=dev); mask = torch.ones(B, 30, dtype=torch.long, device=dev)
y = torch.randint(0, 430, (B,), device=dev)
orig = bench.transformers_encoders_ms
import transformers
# re-run the leg's model but time every step
src = bench.transformers_encoders_ms(dev, vm, imgs, ids, mask, y)
print("leg result (3 warm-up + 10 timed): %.2f ms" % src)
src = bench.transformers_encoders_ms(dev, vm, imgs, ids, mask, y)
print("second call in the same process: %.2f ms" % src)
