#!/bin/bash
# Runs on the GPU box (through gpurun): the default bench line, a rocprofv3 kernel trace of the same command and the
# two PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, no tracing domains beside --kernel-trace) whose summaries
# tools/summarize_profiles.py turns into profiles/<tag>_*.  Everything lands under gpurun_out/prof/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd $R && python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $O/pmc_write_bench.json 2> $O/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $O/pmc_clk_bench.json 2> $O/pmc_clk.err
# keep what travels back small: the per-dispatch CSVs are large
find $O -name '*_agent_info.csv' -delete
ls -la $O $O/*/* | head -40
tail -c 600 $O/bench_n1.json
