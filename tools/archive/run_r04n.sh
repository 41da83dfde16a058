#!/bin/bash
# r04n: config 9 (256x160, one wave per SIMD) for the short-K launches of the 32x32 / 16x16 levels, in situ (variant build, same box)
mkdir -p gpurun_out/r04n
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/cfg9.so
L=$REFACE_HIP_LIB
bash tools/abenv.sh "A=0" "RF_MCFG_M=16384 RF_MCFG_CFG=9 RF_MCFG_KMAX=2560" "A=0" "RF_MCFG_M=16384 RF_MCFG_CFG=9 RF_MCFG_KMAX=640" "RF_MCFG_M=4096 RF_MCFG_CFG=9 RF_MCFG_KMAX=5120" "RF_MCFG_M=65536 RF_MCFG_CFG=9 RF_MCFG_KMAX=1280" "A=0" 2>&1 | tee gpurun_out/r04n/ab.txt
