#!/usr/bin/env python3
"""Per-kernel breakdown of the LAST bench step from a rocprofv3 kernel trace; a step is delimited by the launches of the
first per-step kernel (k_reduce_u64 of the first to_hat after the previous step's last kernel is found by the longest
idle gap pattern is fragile, so: the trace is cut at the launches of `marker` with index multiple of `per_step`).
usage: step_breakdown2.py <dir> <marker substring> <marker launches per step>"""
import collections, csv, glob, os, sys
d, marker, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = max(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
tr = list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
idx = [i for i, r in enumerate(tr) if marker in r["Kernel_Name"]]
seg = tr[idx[-2 * per]:idx[-per]]
agg = collections.OrderedDict()
for r in seg:
    k = r["Kernel_Name"].split("(")[0]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += dur(r)
tot = sum(v[1] for v in agg.values())
print("# kernels=%d busy=%.3f ms wall=%.3f ms" % (len(seg), tot / 1e6, (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-58s launches=%4d %10.1f us %5.1f%%" % (k[:58], v[0], v[1] / 1e3, 100 * v[1] / tot))
