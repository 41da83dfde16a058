# A/B: de-phased DMA issue (spread4: waves 0-3 at k-step 3, waves 4-7 at k-step 0 of the next tile) vs the tree (all waves at k-step 3)
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/spread4.so
python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "linear or conv or geglu or groupnorm_stats or split_k" 2>&1 | tail -2
python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "bench_shapes_bf16 or batch16" 2>&1 | tail -2
for v in "" spread4; do if [ -n "$v" ]; then export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so; else unset REFACE_HIP_LIB; fi; echo "== [$v]"; python tools/bench_gemm.py --only "conv3x3" --reps 20 2>&1 | grep -v amdgpu.ids; python tools/bench_gemm.py --only "geglu" --reps 20 2>&1 | grep -v amdgpu.ids; python tools/bench_gemm.py --only "@64" --reps 20 2>&1 | grep "qkv\|ff2\|proj"; done
bash tools/ab.sh "" spread4 "" spread4
