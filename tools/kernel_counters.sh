#!/bin/bash
# Runs on the GPU box (through gpurun): per-kernel SQ counters of the headline step — three rocprofv3 --pmc passes (one counter group
# each, no tracing domains) over `bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1`, summed per kernel and divided by
# its dispatches.  Writes gpurun_out/pmc/<tag>_kernel_counters.txt (copy it to profiles/).
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $O/p$i -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $O/p$i.json 2> $O/p$i.err
done
python3 - > $O/${TAG}_kernel_counters.txt <<PY
import csv, glob, collections
print("# per-kernel SQ counters of the headline step (whole-GPU sums per dispatch, averaged over the dispatches of a kernel);")
print("# rocprofv3 --pmc <group> -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1, one pass per group (tools/kernel_counters.sh)")
for i in (1, 2, 3):
    f = glob.glob("$O/p%d/**/*counter_collection.csv" % i, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lab::" not in k: continue
        k = k.split("(")[0][:52]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); cnt[k] += 1
    names = sorted({c for k in acc for c in acc[k]})
    print("\n%-54s %6s  %s" % ("kernel", "disp.", "  ".join("%22s" % n for n in names)))
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:16]:
        print("%-54s %6d  %s" % (k, cnt[k], "  ".join("%22.4e" % (acc[k][c] / cnt[k]) for c in names)))
PY
find $O -name '*.csv' -delete
cat $O/${TAG}_kernel_counters.txt
