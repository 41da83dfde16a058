#!/bin/bash
# the round-2 gemm.hip (r02.so) against today's (exp.so) on the residual-carrying short-K launches, cold operands
B="python tools/bench_gemm.py --reps 20 --only"
for v in exp r02 exp r02; do
  export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so
  echo "== $v cold"; for c in "ff2" "proj+res" "geglu 320" "qkv 320"; do $B "$c" --cold 1 2>&1 | grep -v amdgpu.ids | grep -v "@8"; done
done
