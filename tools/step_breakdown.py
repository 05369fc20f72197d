#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py: the kernels between the first step-3 launch of
the second-to-last step and that of the last step.   python tools/step_breakdown.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
big = [i for i, r in enumerate(rows) if "k_gemm_i8_w3<6" in r["Kernel_Name"] and dur(r) > 3e6]
a, b = big[-6], big[-3]
tot = collections.OrderedDict()
for r in rows[a:b]:
    t = tot.setdefault(r["Kernel_Name"].split("(")[0], [0, 0])
    t[0] += 1
    t[1] += dur(r)
busy = sum(v[1] for v in tot.values())
wall = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print("kernels=%d busy=%.3f ms wall=%.3f ms" % (b - a, busy / 1e6, wall / 1e6))
for n, v in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%-55s %3d %9.1f us" % (n[:55], v[0], v[1] / 1e3))
