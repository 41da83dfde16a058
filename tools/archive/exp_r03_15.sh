#!/bin/bash
# upper bound of a persistent tile loop for the multi-round K <= 1280 launches: the same launches with the prologue's wait for the first tile removed
# (RF_GEMM_DBG 128; results wrong), and with no epilogue at all (1)
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
B="python tools/bench_gemm.py --reps 20 --only"
for dbg in 0 128 1 129; do
  echo "== RF_GEMM_DBG=$dbg"
  for c in "geglu" "qkv 320"; do RF_GEMM_DBG=$dbg $B "$c" --cold 1 2>&1 | grep -v amdgpu.ids; done
done
