# A/B: spread variants, and the "pieces never waited for" decomposition
for v in spread spread2; do export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so; for dbg in 0 64; do echo "== $v RF_GEMM_DBG=$dbg"; RF_GEMM_DBG=$dbg python tools/bench_gemm.py --only "conv3x3" --reps 20 2>&1 | grep -v amdgpu.ids; done; done
bash tools/ab.sh spread spread2 spread spread2
