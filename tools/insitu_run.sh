#!/bin/bash
# on the GPU box: in-situ per-layer profile of the current library -> gpurun_out/insitu_<tag>/ ;  tools/insitu_run.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/insitu_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-json gpurun_out/insitu_$tag.prof.json > gpurun_out/insitu_$tag.log 2>&1
tail -1 gpurun_out/insitu_$tag.log | cut -c1-140
python tools/insitu.py $(ls gpurun_out/insitu_$tag/*/*kernel_trace.csv) gpurun_out/insitu_$tag.prof.json --by-shape > gpurun_out/insitu_$tag.txt
head -9 gpurun_out/insitu_$tag.txt
