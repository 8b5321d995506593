"""Prints the kernels of a rocprofv3 kernel trace that ran between the last two marker
launches (a cos_ kernel) (tools/dense_graph_prof.py): kernel, start, gap to the previous kernel's end, duration."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'cos_kernel' in r['Kernel_Name'] or 'cos_' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
step = rows[a + 1:b]
t0 = int(step[0]['Start_Timestamp'])
prev_end, busy = None, 0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = max(prev_end or e, e)
    busy += e - s
    n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:70]
    print(f"{(s - t0) / 1e3:8.1f} +{gap:6.1f} dur={(e - s) / 1e3:7.2f}  {n}")
print('kernels', len(step), 'busy us', busy / 1e3, 'span us', (prev_end - t0) / 1e3)
