"""Prints one steady-state step of a rocprofv3 kernel trace of bench.py (kernel, start, gap, duration)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'dp_adamw_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
print('kernels per step:', len(step))
t0 = int(step[0]['Start_Timestamp'])
tot, prev_end = 0, None
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    prev_end = e
    tot += e - s
    n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:64]
    print(f"{(s - t0) / 1e3:8.1f} +{gap:6.1f} dur={(e - s) / 1e3:7.2f}  {n}")
print('busy us', tot / 1e3, 'span us', (int(step[-1]['End_Timestamp']) - t0) / 1e3)
