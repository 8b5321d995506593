#!/bin/bash
# rocprofv3 kernel trace of a python tool; prints per-kernel statistics.  usage: tools/prof_tool.sh <tag> <script> [args]
set -e
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
script=$root/$1; shift
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 "$script" "$@" > "$out/tool.log" 2>&1
cd "$root"
python3 - "$out/run_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(__import__("os").environ.get("PROF_ROWS", "12"))]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}% calls={r["Calls"]:>5} avg={float(r["AverageNs"])/1e3:8.2f}us min={float(r["MinNs"])/1e3:8.2f} max={float(r["MaxNs"])/1e3:8.2f}  {r["Name"][:80]}')
PY
