#!/bin/bash
# L2 hit rate and MFMA busy counters of one GEMM kind (tools/one_gemm.py).  usage: tools/pmc_gemm.sh <kind> [args]
set -e
export TMPDIR=/tmp
root=$PWD
kind=$1
out=$root/gpurun_out/pmc_gemm_$kind
rm -rf "$out"; mkdir -p "$out"
cd /tmp
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/l2" -o run -- python3 "$root/tools/one_gemm.py" "$@" > "$out/l2.log" 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d "$out/sq" -o run -- python3 "$root/tools/one_gemm.py" "$@" > "$out/sq.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
for sub in ("l2", "sq"):
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k] += 1
        for k, d in acc.items():
            if "gemm" in k or "linear" in k or "wgrad" in k or "Cijk" in k:
                print(sub, k, {c: f"{v:.4g}" for c, v in d.items()})
PY
