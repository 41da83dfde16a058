#!/bin/bash
# what the direct epilogue's 33-55 us (qkv / GEGLU at 64x64) consist of: no stores (8), all stores into one MB (16: L2-resident), no epilogue (1)
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
B="python tools/bench_gemm.py --reps 20 --only"
for dbg in 0 8 16 1; do
  echo "== RF_GEMM_DBG=$dbg"
  for c in "geglu 320" "qkv 320" "proj 320" "ff2 1280->320"; do RF_GEMM_DBG=$dbg $B "$c" --cold 1 2>&1 | grep -v amdgpu.ids; done
done
