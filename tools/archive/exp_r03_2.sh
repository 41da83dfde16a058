# main-loop decomposition of the bf16 GEMM: with / without the LDS-DMA of the main loop (stale operands), with / without epilogue
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
for dbg in 0 32 1 33; do echo "== RF_GEMM_DBG=$dbg"; RF_GEMM_DBG=$dbg python tools/bench_gemm.py --only "conv3x3" --reps 20 2>&1 | grep -v amdgpu.ids; RF_GEMM_DBG=$dbg python tools/bench_gemm.py --only "lin big" --reps 20 2>&1 | grep -v amdgpu.ids; RF_GEMM_DBG=$dbg python tools/bench_gemm.py --only "geglu 640" --reps 20 2>&1 | grep -v amdgpu.ids; done
