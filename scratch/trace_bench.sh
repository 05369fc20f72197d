#!/bin/bash
# usage: bash scratch/trace_bench.sh <tag>  — tests subset, bench and per-step breakdown
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R && timeout -k 10 400 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_prover.py -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -2 $O/tests.log
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2>$O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 > /dev/null 2>&1
cd $R; python tools/step_breakdown.py $O/trace/*/*_kernel_trace.csv | head -${2:-16}; rm -rf $O/trace
