#!/bin/bash
# split-K levels (8x8: M = 1024, 16x16: M = 4096): where the fixed cost goes (epilogue / main loop / reduce pass), and other tile choices
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
B="python tools/bench_gemm.py --reps 30 --only"
for dbg in 0 1 2 3; do echo "== RF_GEMM_DBG=$dbg (1: no epilogue, 2: no main loop)"; RF_GEMM_DBG=$dbg $B "1280 @" 2>&1 | grep -v amdgpu.ids; done
echo "== GPU-side durations (kernel trace)"
bash tools/kt_case.sh "1280 @"
for cfg in 0 1 3 6; do
  echo "== M=1024 -> cfg $cfg"; RF_MCFG_M=1024 RF_MCFG_CFG=$cfg $B "@8" 2>&1 | grep -v amdgpu.ids
  echo "== M=4096 -> cfg $cfg"; RF_MCFG_M=4096 RF_MCFG_CFG=$cfg $B "@16" 2>&1 | grep -v amdgpu.ids
done
echo "== cfg 6 with the 4-stage ring"
RF_GEMM_DEEP=600 RF_MCFG_M=1024 RF_MCFG_CFG=6 $B "@8" 2>&1 | grep -v amdgpu.ids
RF_GEMM_DEEP=600 RF_MCFG_M=4096 RF_MCFG_CFG=6 $B "@16" 2>&1 | grep -v amdgpu.ids
