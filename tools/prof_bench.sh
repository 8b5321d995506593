#!/bin/bash
# rocprofv3 kernel trace of the default benchmark step; prints the per-kernel statistics.
# usage: tools/prof_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
cd $OLDPWD
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}% calls={r["Calls"]:>5} avg={float(r["AverageNs"])/1e3:8.2f}us min={float(r["MinNs"])/1e3:8.2f}  {r["Name"][:110]}')
PY
tail -1 "$out/bench.log" | cut -c1-300
