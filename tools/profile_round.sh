#!/bin/bash
# On the GPU box: the round's evidence for one BASELINE config -- rocprofv3 kernel-trace stats of bench.py, the bench line of that same
# (profiled) run, a clean bench line, and the PMC HBM-traffic passes.   tools/profile_round.sh <tag> <c1|c3|c4>
#   -> gpurun_out/<tag>_<cfg>_{rocprofv3_kernel_stats.csv,bench_under_rocprof.json,bench.json,traffic.json}   (copy to profiles/)
tag=$1; cfg=${2:-c1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${cfg}_rocprof -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs $EXTRA > gpurun_out/${tag}_${cfg}_bench_under_rocprof.json 2> gpurun_out/${tag}_${cfg}_rocprof.log
cp $(ls gpurun_out/${tag}_${cfg}_rocprof/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_${cfg}_rocprofv3_kernel_stats.csv
head -8 gpurun_out/${tag}_${cfg}_rocprofv3_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/${tag}_${cfg}_rocprof
python3 bench.py --config $cfg --steps 3 --warmup 1 --no-other-configs $EXTRA > gpurun_out/${tag}_${cfg}_bench.json 2> gpurun_out/${tag}_${cfg}_bench.log
head -c 400 gpurun_out/${tag}_${cfg}_bench.json; echo
bash tools/pmc_traffic.sh ${tag}_${cfg} $cfg | head -6
rm -rf gpurun_out/${tag}_${cfg}_pmc_FETCH_SIZE gpurun_out/${tag}_${cfg}_pmc_WRITE_SIZE
