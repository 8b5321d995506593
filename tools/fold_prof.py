"""Developer tool: the one-launch bf16 query fold alone (both branches), for counters / traces.  args: [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D, H = 512, 8
dev = torch.device("cuda:0")
lib = _lib.load()
gen = torch.Generator(device=dev).manual_seed(1)
qs = [torch.randn(B, D, device=dev, generator=gen).bfloat16() for _ in range(2)]
ws = [(torch.randn(3 * D, D, device=dev, generator=gen) / D ** 0.5).bfloat16() for _ in range(2)]
bs = [(torch.randn(3 * D, device=dev, generator=gen) * 0.1).bfloat16() for _ in range(2)]
qp = [torch.empty(B, D, device=dev) for _ in range(2)]
u = [torch.empty(B, H, D, device=dev) for _ in range(2)]
folds = (_lib.FoldDesc * 2)(*[_lib.FoldDesc(_lib.ptr(qs[i]), _lib.ptr(ws[i]), _lib.ptr(bs[i]), _lib.ptr(qp[i]), _lib.ptr(u[i]))
                              for i in range(2)])
st = _lib.stream_ptr(dev)
for _ in range(10):
    _lib.check(lib.vqaf_fold_query_multi(ctypes.cast(folds, ctypes.c_void_p), 2, B, D, H, _lib.BF16, _lib.BF16, st), "fold")
torch.cuda.synchronize()
