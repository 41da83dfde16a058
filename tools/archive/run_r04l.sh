#!/bin/bash
mkdir -p gpurun_out/r04l
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04l/prof -- python3 tools/vendor_names.py > gpurun_out/r04l/shapes.txt 2> gpurun_out/r04l/err.txt
cat gpurun_out/r04l/shapes.txt
cp $(ls gpurun_out/r04l/prof/*/*kernel_stats.csv | head -1) gpurun_out/r04l/kernel_stats.csv
rm -rf gpurun_out/r04l/prof
head -30 gpurun_out/r04l/kernel_stats.csv | cut -c1-330
