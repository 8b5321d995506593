#!/bin/bash
# SQ counters of the dense branch's token kernels (tools/dense_stamps.py drives them).  usage: tools/pmc_token_bwd.sh B N
set -e
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc_token_bwd
rm -rf "$out"; mkdir -p "$out"
cd /tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$out/sq" -o run -- python3 "$root/tools/dense_stamps.py" "$@" > "$out/sq.log" 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d "$out/sq2" -o run -- python3 "$root/tools/dense_stamps.py" "$@" > "$out/sq2.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
for sub in ("sq", "sq2"):
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:50]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            if "token_bwd" in k:
                print(sub, k, {c: f"{v / n[(k, c)]:.4g}" for c, v in d.items()})
PY
tail -3 "$out/sq.log" "$out/sq2.log" | cut -c1-300
