#!/bin/bash
# the 8x8 level with cold operands (every launch behind a pass over 1 GB): default tiles, the ring, other tile shapes
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
B="python tools/bench_gemm.py --reps 15 --only @8"
echo "== warm"; $B 2>&1 | grep -v amdgpu.ids
echo "== cold"; $B --cold 1 2>&1 | grep -v amdgpu.ids
echo "== cold, ring off"; RF_GEMM_DEEP=0 $B --cold 1 2>&1 | grep -v amdgpu.ids
for cfg in 0 6 4 5; do echo "== cold, M=1024 -> cfg $cfg (ring <= 256)"; RF_GEMM_DEEP=256 RF_MCFG_M=1024 RF_MCFG_CFG=$cfg $B --cold 1 2>&1 | grep -v amdgpu.ids; done
echo "== cold, no main loop / no epilogue"; for dbg in 1 2 3; do RF_GEMM_DBG=$dbg $B --cold 1 2>&1 | grep -v amdgpu.ids; done
echo "== cold @16"; python tools/bench_gemm.py --reps 15 --only "@16" --cold 1 2>&1 | grep -v amdgpu.ids
