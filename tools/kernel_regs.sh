#!/bin/bash
# Diagnostics: registers / spills of ONE conv_gemm_kernel instantiation (seconds instead of the 5 minutes of the whole unit):
#   tools/kernel_regs.sh "unsigned short, unsigned short, 4, 2, 2, 5, true, true, 2, 1, false, 0, true"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cat > $tmp/one.hip <<EOF
#define RF_KERNEL_ONLY
#include "$ROOT/reface_amd/csrc/gemm.hip"
template __global__ void rf::conv_gemm_kernel<$1>(const rf::GemmParams);
EOF
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$ROOT/reface_amd/csrc --cuda-device-only -S -o $tmp/one.s $tmp/one.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|Spill|ScratchSize|SGPRs:|Occupancy" | head -8
[ -n "$2" ] && cp $tmp/one.s $2
rm -rf $tmp
