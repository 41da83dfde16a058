#!/bin/bash
# r04i: numerics of fp16 / bf16 single-plane A operands (tools/build_a16_variants.sh) at the full-S full-width gate
mkdir -p gpurun_out/r04i
for v in a16 abf16; do
  REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "(ddim50 or ddim5) and f32x3" > gpurun_out/r04i/$v.log 2>&1
  grep -E "CFG DDIM|passed|failed|assert" gpurun_out/r04i/$v.log | head -12
done
