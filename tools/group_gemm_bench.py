"""Times the grouped f32 product launches of the dense branch's backward, whole and by subsets of their
problems (developer tool): which problem sets the duration of a launch."""
import math
import sys

import torch

sys.path.insert(0, '/root/repo')
from vqa_amd.xattn import _gemm_group, _ones, _prob  # noqa: E402

dev = torch.device('cuda:0')
B, E, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 128), 512, 8
dh = E // H
nb = 2
bf = dict(dtype=torch.bfloat16, device=dev)
f32 = dict(dtype=torch.float32, device=dev)
g = torch.randn(B, nb * E, **bf)
w_in = [torch.randn(3 * E, E, **bf) for _ in range(nb)]
attn = torch.randn(nb, B, E, **f32)
z = torch.randn(nb, B, H, E, **f32)
d_attn = torch.randn(nb, B, E, **f32)
t = torch.zeros(nb, B, H, E, **f32)
d_wout = torch.zeros(nb, E, E, **f32)
d_win = torch.zeros(nb, 3 * E, E, **f32)
d_bout = torch.zeros(nb, E, **f32)
d_bin = torch.zeros(nb, 3 * E, **f32)
du = torch.randn(nb, B, H, E, **f32)
qp = torch.randn(nb, B, E, **f32)
d_qp = torch.zeros(nb, B, E, **f32)
ones = (_ones(dev), 0)


def problems(which):
    out = []
    for i in range(nb):
        if 't' in which:
            out.append(_prob((d_attn[i], 0), (dh, E, 1), (w_in[i], 2 * E * E), (dh * E, E, 1), (t[i], 0), (E, H * E), H, B, E, dh, store=True))
        if 'wout' in which:
            out.append(_prob((g, i * E), (0, 1, nb * E), (attn[i], 0), (0, E, 1), (d_wout[i], 0), (0, E), 1, E, E, B, store=True))
        if 'wv' in which:
            out.append(_prob((d_attn[i], 0), (dh, 1, E), (z[i], 0), (E, H * E, 1), (d_win[i], 2 * E * E), (dh * E, E), H, dh, E, B,
                             store=True))
        if 'bout' in which:
            out.append(_prob(ones, (0, 0, 0), (g, i * E), (0, nb * E, 1), (d_bout[i], 0), (0, E), 1, 1, E, B))
        if 'bin' in which:
            out.append(_prob(ones, (0, 0, 0), (d_attn[i], 0), (0, E, 1), (d_bin[i], 2 * E), (0, E), 1, 1, E, B))
        if 'dqp' in which:
            out.append(_prob((du[i], 0), (E, H * E, 1), (w_in[i], E * E), (dh * E, 1, E), (d_qp[i], 0), (dh, E), H, B, dh, E,
                             alpha=1 / math.sqrt(dh)))
        if 'wk' in which:
            out.append(_prob((qp[i], 0), (dh, 1, E), (du[i], 0), (E, H * E, 1), (d_win[i], E * E), (dh * E, E), H, dh, E, B,
                             store=True))
    return out


def timed(which, reps=200):
    probs = problems(which)
    for _ in range(10):
        _gemm_group(probs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _gemm_group(probs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for which in (('t',), ('wout',), ('wv',), ('bout',), ('bin',), ('t', 'wout', 'wv'), ('t', 'wout', 'wv', 'bout', 'bin'),
              ('dqp',), ('wk',), ('dqp', 'wk')):
    print(f"B={B} {'+'.join(which):24s} {timed(which):7.2f} us", flush=True)
