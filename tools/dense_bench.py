"""Developer tool: the dense (unpruned) CoAttention leg of bench.py, both launch forms side by side
(branches batched into the same launches / one chain of launches per branch on two streams)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqa_amd import fusion  # noqa: E402

dev = torch.device("cuda:0")
init = fusion.CoAttention.__init__
for batched in (True, False):
    def patched(self, *a, _b=batched, **k):
        init(self, *a, **k)
        self.batch_branches = _b
    fusion.CoAttention.__init__ = patched
    print("batched" if batched else "per-branch", json.dumps(bench.dense_leg(dev, torch.bfloat16)))
fusion.CoAttention.__init__ = init
