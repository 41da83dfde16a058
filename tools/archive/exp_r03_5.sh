# A/B: thirds-spread (spread3) vs k-step-3 spread (spread); correctness of spread3 first
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/spread3.so
python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "linear or conv or geglu or groupnorm_stats or split_k" 2>&1 | tail -3
python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "bench_shapes_bf16 or batch16" 2>&1 | tail -3
for v in spread spread3; do export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so; for dbg in 0 64; do echo "== $v RF_GEMM_DBG=$dbg"; RF_GEMM_DBG=$dbg python tools/bench_gemm.py --only "conv3x3" --reps 20 2>&1 | grep -v amdgpu.ids;  RF_GEMM_DBG=$dbg python tools/bench_gemm.py --only "geglu" --reps 20 2>&1 | grep -v amdgpu.ids; done; done
bash tools/ab.sh spread spread3 spread spread3
