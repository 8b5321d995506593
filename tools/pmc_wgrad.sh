#!/bin/bash
# Counter passes over tools/wgrad_one.py (one rocprofv3 run per counter group), per-launch means of the GEMM kernel.
set -e
export TMPDIR=/tmp
root=$PWD
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_wgrad_$i
  rm -rf "$out"; mkdir -p "$out"
  cd /tmp
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out" -o run -- python3 "$root/tools/wgrad_one.py" "$@" > "$out/tool.log" 2>&1
  cd "$root"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_wgrad_*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-32s mean per launch %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
