#!/bin/bash
# Upper bound of a row-extended A tile shared by the three horizontal taps of a 3x3 window (timing only, wrong results): RF_GEMM_DBG bit 8 drops the
# A pieces of two K tiles out of three in the stride-1 3x3 convolutions.  Same box, interleaved.
cd $GRAFT_REPO_ROOT
L=$PWD/reface_amd/lib/alt/base.so
bash tools/abenv.sh "REFACE_HIP_LIB=$L RF_GEMM_DBG=0" "REFACE_HIP_LIB=$L RF_GEMM_DBG=256" "REFACE_HIP_LIB=$L RF_GEMM_DBG=0" "REFACE_HIP_LIB=$L RF_GEMM_DBG=256"
