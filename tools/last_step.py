#!/usr/bin/env python3
"""Per-kernel time of the LAST bench step in a rocprofv3 kernel trace (steps end with k_scatter_jobs)."""
import csv, glob, os, sys, collections
d0 = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof64'
f = max(glob.glob(d0 + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
tr = list(csv.DictReader(open(f))); tr.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(tr) if 'k_scatter_jobs' in r['Kernel_Name']]
last = tr[ends[-2] + 1: ends[-1] + 1]
d = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r['Kernel_Name'].split('(')[0].replace('lab::', '').replace('void ', '')
    d[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; d[k][1] += 1
tot = sum(v[0] for v in d.values())
print(len(last), 'launches, busy %.1f ms, wall %.1f' % (tot / 1e3, (int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e6))
for k, v in sorted(d.items(), key=lambda kv: -kv[1][0]): print('%-40s %8.0f us x%d  %4.1f%%' % (k[:40], v[0], v[1], 100 * v[0] / tot))
if len(sys.argv) > 2:
    for r in last:
        if sys.argv[2] in r['Kernel_Name']:
            print(r['Kernel_Name'].split('(')[0][-24:], '%.0f us' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
