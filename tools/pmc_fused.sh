#!/bin/bash
# HBM-side traffic of the one-launch pruning branch from the L2's fabric counters: FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 passes over tools/fused_bench.py (B = 1024, 197 + 30 tokens, bf16).
set -e
export TMPDIR=/tmp
root=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_fused_$c
  rm -rf "$out"; mkdir -p "$out"
  cd /tmp
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out" -o run -- python3 "$root/tools/fused_bench.py" > "$out/tool.log" 2>&1
  cd "$root"
done
python3 - <<'PY'
import csv, json, glob
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_fused_{c}/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "fused_prune_pool" in r["Kernel_Name"] and r["Counter_Name"] == c]
    vals = [float(r["Counter_Value"]) for r in rows]
    # tools/fused_bench.py launches 24 x vision only, 24 x text only, 24 x both
    res[c] = {"vision": vals[4:24], "text": vals[28:48], "both": vals[52:72]}
out = {"shape": "B=1024, 197 vision + 30 text tokens, D=512, bf16 (tools/fused_bench.py)",
       "correction": "gfx950: FETCH_SIZE counts a wide coalesced read at half its bytes -> doubled; WRITE_SIZE exact; "
                     "both are L2 fabric-side counters, so reads served by the Infinity Cache are included",
       "kernels": {}}
for k in ("vision", "text", "both"):
    fk = sum(res["FETCH_SIZE"][k]) / len(res["FETCH_SIZE"][k])
    wk = sum(res["WRITE_SIZE"][k]) / len(res["WRITE_SIZE"][k])
    out["kernels"][f"fused_prune_pool_kernel<16> ({k})"] = {
        "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
        "fabric_bytes_per_launch": int(2 * fk * 1024 + wk * 1024), "launches": len(res["FETCH_SIZE"][k])}
json.dump(out, open("gpurun_out/r02_pmc_fused.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
