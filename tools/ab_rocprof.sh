#!/bin/bash
# On the GPU box: kernel-trace stats of a short bench run under an environment switch.   tools/ab_rocprof.sh <tag> VAR=value ...
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_${tag} -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-roofline > /dev/null 2> gpurun_out/ab_${tag}.log
cp $(ls gpurun_out/ab_${tag}/*/*kernel_stats.csv | head -1) gpurun_out/ab_${tag}_kernel_stats.csv
rm -rf gpurun_out/ab_${tag}
