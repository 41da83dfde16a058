#!/bin/bash
# r05s: configs[4] (every weight fp8) with the 2-D patch tile order restricted to bf16 / fp32 operands (the in-tree library) against the build that applied it to the fp8 kernels too (alt/patchall.so = the library of the first r05m evidence set)
mkdir -p gpurun_out/r05s
F="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs"
one() { if [ -n "$1" ]; then export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$1.so; else unset REFACE_HIP_LIB; fi
  python bench.py $F $2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib [%-9s] %-24s %.1f ms/batch  %.3f img/s' % ('$1', '$2', r['ms_per_step'], r['value']))"; }
{
for i in 1 2; do one patchall "--config c4"; one "" "--config c4"; done
one patchall "--config c4 --dtype fp8w"; one "" "--config c4 --dtype fp8w"
one patchall ""; one "" ""; one patchall ""; one "" ""
} | tee gpurun_out/r05s/ab.txt
