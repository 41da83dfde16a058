#!/bin/bash
# r05u: fused feed-forward kernel -- A-fragment (W2 / Wpo) LDS reads 1 / 2 / 3 / 4 MFMAs ahead in GEMM 2 / 3 (RF_AF_DEPTH), same-box A/B on the whole bench
out=gpurun_out/r05u; mkdir -p $out
tools/ab.sh afd1 afd3 afd2 afd4 afd1 afd3 afd2 afd4 2>&1 | tee $out/ab.txt
