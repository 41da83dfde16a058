#!/bin/bash
# r04k: our kernels against PyTorch-ROCm's vendor libraries (hipBLASLt / MIOpen / SDPA / native group_norm) on the UNet's shapes, same box, same tensors
mkdir -p gpurun_out/r04k
timeout 900 python tools/bench_gemm.py --vendor 1 --reps 20 --json gpurun_out/r04k/vendor_default.json > gpurun_out/r04k/vendor_default.txt 2>&1
tail -60 gpurun_out/r04k/vendor_default.txt | cut -c1-200
timeout 900 python tools/bench_gemm.py --vendor 2 --reps 20 --only conv --json gpurun_out/r04k/vendor_miopen_find.json > gpurun_out/r04k/vendor_miopen_find.txt 2>&1
tail -14 gpurun_out/r04k/vendor_miopen_find.txt | cut -c1-200
