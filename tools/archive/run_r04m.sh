#!/bin/bash
# r04m: 256x160 tile, 4 waves (one per SIMD), wave tile 64x160 -- experiment config 9 of a variant build
mkdir -p gpurun_out/r04m
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/cfg9.so
for only in "@32" "@16" "@64"; do
  echo "== base $only"; python tools/bench_gemm.py --only "$only" --reps 30 2>/dev/null | grep -v "attn\|gn \|ln " 
  echo "== cfg9 $only"; RF_GEMM_CFG=9 python tools/bench_gemm.py --only "$only" --reps 30 2>/dev/null | grep -v "attn\|gn \|ln "
done > gpurun_out/r04m/bench.txt 2>&1
cat gpurun_out/r04m/bench.txt
echo "== cold"; for c in "" 9; do RF_GEMM_CFG=$c python tools/bench_gemm.py --only "640 @32" --cold 1 --reps 20 2>/dev/null; done | tee gpurun_out/r04m/cold.txt
RF_GEMM_CFG=9 timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv or linear or gemm or splitk or stats" 2>&1 | tail -8 | tee gpurun_out/r04m/pytest.txt
