#!/bin/bash
# Memory-fabric traffic of the one-launch pruning branch from the L2's fabric counters: FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 passes over tools/fused_bench.py (B = 1024, 197 + 30 tokens, bf16),
# for the two-pass kernel and (ONEPASS=1 in the tool's environment) the one-pass kernel.
# Usage: tools/pmc_fused.sh <tag>     -> gpurun_out/<tag>_pmc_fused.json
set -e
export TMPDIR=/tmp
root=$PWD
tag=${1:-r03}
for mode in twopass onepass; do
  for c in FETCH_SIZE WRITE_SIZE; do
    out=$root/gpurun_out/pmc_fused_${mode}_$c
    rm -rf "$out"; mkdir -p "$out"
    cd /tmp
    if [ $mode = onepass ]; then export ONEPASS=1; else unset ONEPASS; fi
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out" -o run -- python3 "$root/tools/fused_bench.py" > "$out/tool.log" 2>&1
    cd "$root"
  done
done
unset ONEPASS
python3 - "$tag" <<'PY'
import csv, json, glob, sys
tag = sys.argv[1]
out = {"shape": "B=1024, 197 vision + 30 text tokens, D=512, bf16 (tools/fused_bench.py)",
       "correction": "gfx950: FETCH_SIZE counts a wide coalesced read at half its bytes -> doubled; WRITE_SIZE exact; "
                     "both are L2 fabric-side counters, so reads served by the Infinity Cache are included",
       "algorithmic_bytes": {"one_pass": 1024 * (197 + 30) * 1024 + 1024 * 2 * (8 * 512 * 4) + 1024 * (98 + 15) * 8 + 1024 * 1024 * 2,
                             "two_pass": 512630784},
       "kernels": {}}
for mode, pat in (("twopass", "fused_prune_pool_kernel"), ("onepass", "fused_onepass_kernel")):
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/pmc_fused_{mode}_{c}/*counter_collection.csv")[0]
        rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"] and r["Counter_Name"] == c]
        vals = [float(r["Counter_Value"]) for r in rows]
        # tools/fused_bench.py launches 24 x vision only, 24 x text only, 24 x both
        res[c] = {"vision": vals[4:24], "text": vals[28:48], "both": vals[52:72]}
    for k in ("vision", "text", "both"):
        fk = sum(res["FETCH_SIZE"][k]) / len(res["FETCH_SIZE"][k])
        wk = sum(res["WRITE_SIZE"][k]) / len(res["WRITE_SIZE"][k])
        out["kernels"][f"{pat} ({k})"] = {
            "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
            "fabric_bytes_per_launch": int(2 * fk * 1024 + wk * 1024), "launches": len(res["FETCH_SIZE"][k])}
json.dump(out, open(f"gpurun_out/{tag}_pmc_fused.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
