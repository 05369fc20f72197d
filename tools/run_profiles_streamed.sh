#!/bin/bash
# Runs on the GPU box (through gpurun): the evidence behind the streamed-B records (profiles/rNN_streamed_*, rNN_crs_expand_valu.json,
# rNN_mfma_ceiling.json).  Everything lands under gpurun_out/prof_s/; tools/crs_expand_valu.py and a copy turn it into profiles/.
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_s
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) kernel trace of one streamed step at kappa = 2048 (same code path as kappa = 2^14: K chunks reduced mod q, 512-row slabs)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --kappa 2048 --crs streamed --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/streamed_k2048_bench.json 2> $O/trace.err
cp $(ls -t $O/trace/*/*_kernel_stats.csv | head -1) $O/${TAG}_streamed_kernel_stats.csv
rm -rf $O/trace
# (2) the expansion kernel alone: rate without the profiler, then vector instructions per coefficient (own --pmc pass)
cd $R && python3 tools/crs_expand_rate.py 5 > $O/rate.json 2> $O/rate.err
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $O/pmc -- python3 $R/tools/crs_expand_rate.py 2 > $O/pmc_rate.json 2> $O/pmc.err
cd $R && python3 tools/crs_expand_valu.py $O/pmc $O/rate.json $TAG && cp profiles/${TAG}_crs_expand_valu.json $O/
rm -rf $O/pmc
# (3) the matrix-core ceiling microbenchmark
(cd $R && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_i8.hip -o tools/ubench_mfma_i8 && tools/ubench_mfma_i8 > $O/${TAG}_mfma_ceiling.json)
ls -la $O
tail -c 400 $O/streamed_k2048_bench.json
