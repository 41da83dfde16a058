#!/bin/bash
# r04ab: pixels per block (RF_GN_PPB; 16 = shipped) and total-block cap (RF_GN_CAP; 1024 = shipped) of the GroupNorm-apply launches (variant build)
mkdir -p gpurun_out/r04ab
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/gnppb.so
bash tools/abenv.sh "RF_GN_PPB=16" "RF_GN_PPB=2" "RF_GN_PPB=1" "RF_GN_PPB=2 RF_GN_CAP=2048" "RF_GN_PPB=2 RF_GN_CAP=4096" "RF_GN_PPB=1 RF_GN_CAP=4096" "RF_GN_PPB=16" "RF_GN_PPB=2" "RF_GN_PPB=1" "RF_GN_PPB=2 RF_GN_CAP=2048" "RF_GN_PPB=2 RF_GN_CAP=4096" "RF_GN_PPB=1 RF_GN_CAP=4096" 2>&1 | tee gpurun_out/r04ab/ab2.txt
