"""Prints the parts of a bench.py JSON line a developer looks at first."""
import json
import sys

d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "launch", "ms_per_step_eager", "n_gpus")})
r = d["roofline"]
print("roofline:", {k: r.get(k) for k in ("kernel_ms", "achieved", "frac", "one_pass_GBs", "traffic", "copy_rate_same_run_GBs")})
for k in ("mfma", "f32", "prune_mix_cfg4", "dense_branch", "full_model_cfg2", "cpu_baseline", "one_pass_kernel"):
    if k in d:
        v = d[k]
        if k == "f32":
            v = {kk: v[kk] for kk in ("value", "ms_per_step")}
        print(k + ":", json.dumps(v)[:600])
