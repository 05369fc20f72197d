#!/usr/bin/env python3
"""Turns the rocprofv3 outputs under gpurun_out/prof/ (kernel trace + two PMC passes of bench.py) into the small
summaries committed under profiles/ (round tag as argv[1], default r01)."""
import collections
import os
import csv
import glob
import json
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = "gpurun_out/prof/"
# the dominant kernel is the one bench.py's profiling hook reports ("lab::<name> — ...")
DOM = sys.argv[2] if len(sys.argv) > 2 else json.load(open(P + "bench_n1.json"))["roofline"]["kernel"].split(" — ")[0].replace("lab::", "")

def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)   # gpurun merges into gpurun_out/: older runs may still be there


shutil.copy(newest(P + "trace/*/*_kernel_stats.csv"), "profiles/%s_rocprofv3_kernel_stats.csv" % tag)
shutil.copy(P + "bench_n1.json", "profiles/%s_bench_n1.json" % tag)
bench = json.load(open(P + "pmc_fetch_bench.json"))
cfg = bench["config"]


def biggest(kind, name):
    f = newest(P + "pmc_%s/*/*_counter_collection.csv" % kind)
    best = None
    for r in csv.DictReader(open(f)):
        if DOM in r["Kernel_Name"] and r["Counter_Name"] == name:
            v = float(r["Counter_Value"])
            if best is None or v > best[0]:
                best = (v, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return best


fs, ws = biggest("fetch", "FETCH_SIZE"), biggest("write", "WRITE_SIZE")
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
fetch_bytes, write_bytes = fs[0] * 1024 * 2, ws[0] * 1024
out = {"kernel": "lab::" + DOM + " (step-3 launch of bench.py: the outer commitment of all proofs of the step, nmat = r*t1 matrices of kappa1 x kappa; one limb of the CRS basis per launch on the matrix-core path)",
       "workload_key": [cfg["q"], cfg["n"], cfg["r"], cfg["kappa"], cfg["nlimbs"], cfg["crs_basis"] != "main basis", cfg["proofs_per_step_per_gpu"]],
       "digit_primes": cfg.get("digit_primes"),
       "step3": "mfma" if cfg.get("step3", "").startswith("int8") else "valu",
       "FETCH_SIZE_KB_raw": fs[0], "WRITE_SIZE_KB_raw": ws[0],
       "correction": "FETCH_SIZE x2 on gfx950 for 16-B/lane coalesced streaming loads (MI355X_MICROARCH.md 'HBM'); "
                     "WRITE_SIZE exact; FETCH and WRITE collected in separate --pmc passes",
       "fetch_bytes": fetch_bytes, "write_bytes": write_bytes, "hbm_bytes_per_launch": fetch_bytes + write_bytes,
       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch_bytes + write_bytes) / alg,
       "kernel_ns_under_pmc": fs[1],
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline "
                  "--no-extras --steps 2 --warmup 1  (tools/run_profiles.sh)"}
# clock the chip holds under the dominant launch: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration (MI355X_MICROARCH.md 'DVFS give-back')
try:
    f = newest(P + "pmc_clk/*/*_counter_collection.csv")
    clk = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if dur > 200000:      # the quotient reads high on short dispatches
                clk[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]) / 8.0 / dur, dur))
    out["clock_GHz_by_kernel"] = {k: {"launches": len(v), "avg_ms": sum(d for _, d in v) / len(v) / 1e6,
                                      "GHz_min": min(c for c, _ in v), "GHz_avg": sum(c for c, _ in v) / len(v), "GHz_max": max(c for c, _ in v)}
                                  for k, v in sorted(clk.items(), key=lambda kv: -sum(d for _, d in kv[1]))}
    out["clock_method"] = "rocprofv3 --pmc GRBM_GUI_ACTIVE (own pass): counter / 8 XCDs / dispatch duration, dispatches longer than 0.2 ms"
except Exception as e:      # pass not run
    out["clock_GHz_by_kernel"] = None
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)

tr = list(csv.DictReader(open(newest(P + "trace/*/*_kernel_trace.csv"))))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
# the Ajtai product of the batch runs on the same kernel template: the step-3 launches are the long ones
dom_durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr if DOM in r["Kernel_Name"]]
big = [i for i, r in enumerate(tr) if DOM in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 0.6 * max(dom_durs)]
tb = json.load(open(P + "trace_bench.json"))
lines = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras  (%s)" % tag,
         "# workload: " + tb["config"]["workload"] + " | CRS: " + tb["config"]["crs_basis"],
         "# bench line under the profiler: " + json.dumps({k: tb[k] for k in ("value", "ms_per_step", "roofline")}),
         "# kernels of ONE timed step (between two consecutive step-3 mat-vec launches), from the kernel trace:"]
per_step = max(1, len(big) // (tb["steps"] + tb["warmup"]))   # step-3 launches per bench step (one per limb and pass over B)
seg = tr[big[-2 * per_step]:big[-per_step]]
agg = collections.OrderedDict()
for r in seg:
    k = r["Kernel_Name"].split("(")[0]
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
lines.append("# kernels=%d  busy=%.3f ms  wall=%.3f ms" % (len(seg), tot / 1e6,
             (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("%-52s launches=%3d  %10.1f us  %5.1f%%" % (k, v[0], v[1] / 1e3, 100 * v[1] / tot))
durs = [(int(tr[i]["End_Timestamp"]) - int(tr[i]["Start_Timestamp"])) / 1e6 for i in big]
lines.append("# step-3 %s launch durations (ms): " % DOM + ", ".join("%.3f" % d for d in durs) + "  avg=%.3f" % (sum(durs) / len(durs)))
open("profiles/%s_step_breakdown.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[3:]))
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}))
print(json.dumps({k: v for k, v in json.load(open(P + "bench_n1.json")).items() if k in ("value", "ms_per_step", "roofline", "cpu_baseline")})[:1500])
