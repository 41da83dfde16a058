#!/bin/bash
# r06n: d = 80 pipelined attention with the prologue reordered (tile 0, Q, tile 1) and 16-byte output stores: tests, phase stamps, isolated time, bench A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06n; O=gpurun_out/r06n
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "attention" > $O/pytest_attention.log 2>&1; tail -3 $O/pytest_attention.log
REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/a80stamp.so python tools/archive/attn_stamp.py 80 1024 2>&1 | tail -6 | tee $O/stamps_d80.txt
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/a80.so
for v in 0 1 0 1; do echo "RF_ATTN_PIPE80=$v"; RF_ATTN_PIPE80=$v python tools/bench_gemm.py --only "attn d80" --cold 1 --reps 30 2>/dev/null | tail -1; RF_ATTN_PIPE80=$v python tools/bench_gemm.py --only "attn d80" --reps 50 2>/dev/null | tail -1; done | tee $O/attn_d80_isolated.txt
bash tools/abenv.sh "RF_ATTN_PIPE80=0" "RF_ATTN_PIPE80=1" "RF_ATTN_PIPE80=0" "RF_ATTN_PIPE80=1" | tee $O/ab_c1.txt
