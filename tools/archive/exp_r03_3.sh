# A/B: LDS-DMA pieces spread between the MFMA columns of k-step 3 (variant "spread") vs one burst behind the barrier (variant "exp" = the tree)
for v in exp spread exp spread; do export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so; echo "== $v"; python tools/bench_gemm.py --only "conv3x3" --reps 20 2>&1 | grep -v amdgpu.ids; python tools/bench_gemm.py --only "geglu" --reps 20 2>&1 | grep -v amdgpu.ids; python tools/bench_gemm.py --only "qkv" --reps 20 2>&1 | grep -v amdgpu.ids; done
bash tools/ab.sh exp spread exp spread
