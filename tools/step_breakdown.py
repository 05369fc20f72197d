#!/usr/bin/env python3
"""Per-kernel breakdown of ONE timed bench step from a rocprofv3 kernel trace.
usage: step_breakdown.py <dir-with-*_kernel_trace.csv> <substring of the step-3 kernel name> [min_ns]"""
import collections, csv, glob, sys
d, dom = sys.argv[1], sys.argv[2]
min_ns = float(sys.argv[3]) if len(sys.argv) > 3 else 3e6
tr = list(csv.DictReader(open(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
big = [i for i, r in enumerate(tr) if dom in r["Kernel_Name"] and dur(r) > min_ns]
seg = tr[big[-2]:big[-1]]
agg = collections.OrderedDict()
for r in seg:
    k = r["Kernel_Name"].split("(")[0]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += dur(r)
tot = sum(v[1] for v in agg.values())
wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print("# kernels=%d  busy=%.3f ms  wall(first start..last end)=%.3f ms" % (len(seg), tot / 1e6, wall / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s launches=%3d  %10.1f us  %5.1f%%" % (k[:60], v[0], v[1] / 1e3, 100 * v[1] / tot))
durs = [dur(tr[i]) / 1e6 for i in big]
print("# step-3 launch durations (ms): " + ", ".join("%.3f" % x for x in durs))
