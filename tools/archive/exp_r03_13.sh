#!/bin/bash
# residual segments of the direct epilogue: more blocks in flight (RF_EPI_PFD 6 / 8) or an L2 touch pass at kernel start (RF_RES_TOUCH)
B="python tools/bench_gemm.py --reps 20 --only"
for v in "" pfd6 pfd8 touch; do
  if [ -n "$v" ]; then export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so; else unset REFACE_HIP_LIB; fi
  echo "== variant [$v] warm"; $B "proj" 2>&1 | grep -v amdgpu.ids; $B "ff2+res" 2>&1 | grep -v amdgpu.ids
  echo "== variant [$v] cold"; $B "proj" --cold 1 2>&1 | grep -v amdgpu.ids; $B "ff2+res" --cold 1 2>&1 | grep -v amdgpu.ids
done
bash tools/ab.sh "" pfd6 pfd8 touch "" pfd6 pfd8 touch
