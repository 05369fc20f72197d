#!/usr/bin/env python3
"""profiles/<tag>_crs_expand_valu.json from a `rocprofv3 --pmc SQ_INSTS_VALU` pass over tools/crs_expand_rate.py:
vector instructions (wave level) per 64 coefficients = vector instructions per coefficient and lane.
usage: crs_expand_valu.py <rocprof output dir> <rate json line file> <tag>"""
import csv
import glob
import json
import sys

d, ratefile, tag = sys.argv[1], sys.argv[2], sys.argv[3]
rate = json.loads([l for l in open(ratefile) if l.startswith("{")][-1])
vals = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_crs_expand" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            vals.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
per_launch = sum(v for v, _ in vals) / len(vals)
coefs = rate["coefficients_per_launch"]
out = {"kernel": "lab::k_crs_expand", "q": rate["q"], "coefficients_per_launch": coefs, "launches_counted": len(vals),
       "SQ_INSTS_VALU_per_launch": per_launch,
       "valu_instructions_per_coefficient_lane": per_launch * 64.0 / coefs,
       "kernel_ms_under_pmc": sum(t for _, t in vals) / len(vals) / 1e6,
       "standalone_rate_without_profiler": rate,
       "valu_peak_lane_ops_per_s": 256 * 4 * 16 * 2.4e9,
       "note": "SQ_INSTS_VALU counts wave-level vector instructions; x 64 lanes / coefficients = instructions per coefficient and lane. "
               "achieved lane operations per second = coefficients/s x that; peak = 256 CUs x 4 SIMD16 x 2.4 GHz",
       "command": "rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -- python3 tools/crs_expand_rate.py"}
out["standalone_frac_of_valu_peak"] = rate["coefficients_per_s"] * out["valu_instructions_per_coefficient_lane"] / out["valu_peak_lane_ops_per_s"]
json.dump(out, open("profiles/%s_crs_expand_valu.json" % tag, "w"), indent=1)
print(json.dumps(out))
