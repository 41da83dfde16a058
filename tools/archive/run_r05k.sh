#!/bin/bash
# r05k: where do the K <= 1280 projections spend their time?  GPU-side durations (rocprofv3) of two shapes under the timing-decomposition switches of an
# RF_EXPERIMENT build (hot operands: the floor of what the step sees)
out=gpurun_out/r05k; mkdir -p $out
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
for shape in "proj+res 1280->1280 @16" "proj+res 640->640 @32" "proj+res 320->320 @64" "qkv 320->960 @64"; do
  echo "=== $shape" | tee -a $out/decomp.txt
  for v in "X=0" "RF_GEMM_DBG=1" "RF_GEMM_DBG=2" "RF_GEMM_DBG=32" "RF_GEMM_DBG=128" "RF_GEMM_DBG=8" "RF_GEMM_DEEP=0" "RF_GEMM_CFG=2" "RF_GEMM_CFG=0" "RF_GEMM_CFG=4" "RF_GEMM_CFG=3"; do
    echo "--- $v" >> $out/decomp.txt
    tools/kt_case.sh "$shape" "$v" >> $out/decomp.txt 2>&1
  done
done
cat $out/decomp.txt
