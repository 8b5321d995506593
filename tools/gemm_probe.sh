#!/bin/bash
# Runs every tools/bin/gemm_probe_* binary under rocprofv3 and prints the per-kernel averages.
# usage: tools/gemm_probe.sh [B]
set -e
export TMPDIR=/tmp
root=$PWD
cd /tmp
for bin in "$root"/tools/bin/gemm_probe_*; do
  tag=$(basename "$bin")
  out=$root/gpurun_out/$tag
  rm -rf "$out"; mkdir -p "$out"
  timeout -k 10 60 "$bin" "${1:-1024}"
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- "$bin" "${1:-1024}" > "$out/log.txt" 2>&1
  f=$(find "$out" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'  {sys.argv[2]} avg={float(r["AverageNs"])/1e3:8.2f}us min={float(r["MinNs"])/1e3:8.2f}  {r["Name"][:70]}')
PY
done
