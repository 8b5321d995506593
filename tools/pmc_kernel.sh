#!/bin/bash
# SQ counters (two passes) of the kernels whose name contains <substring>, while <script> runs.
# usage: tools/pmc_kernel.sh <substring> <script> [args]
set -e
export TMPDIR=/tmp
root=$PWD
sub=$1; shift
script=$root/$1; shift
out=$root/gpurun_out/pmc_$sub
rm -rf "$out"; mkdir -p "$out"
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$out/sq" -o run -- python3 "$script" "$@" > "$out/sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVES --output-format csv -d "$out/sq2" -o run -- python3 "$script" "$@" > "$out/sq2.log" 2>&1
cd "$root"
python3 - "$out" "$sub" <<'PY'
import csv, glob, sys, collections
for sub in ("sq", "sq2"):
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            if sys.argv[2] in k:
                print(sub, k, {c: f"{v / n[(k, c)]:.4g}" for c, v in d.items()})
PY
