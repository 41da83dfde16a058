#!/bin/bash
# r06r: the attention unit restructured (XS: max | all of a unit's MFMAs beside all of its exponentials): tests with XS forced on, stamps, isolated times, bench A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06r; O=gpurun_out/r06r
L=$PWD/reface_amd/lib/alt
REFACE_HIP_LIB=$L/xs.so RF_ATTN_XS40=1 RF_ATTN_XS80=1 timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "attention" > $O/pytest_attention_xs.log 2>&1; tail -3 $O/pytest_attention_xs.log
for x in 0 1; do echo "XS=$x"; REFACE_HIP_LIB=$L/xsstamp.so RF_ATTN_XS40=$x RF_ATTN_XS80=$x python tools/archive/attn_stamp.py 40 4096 2>&1 | tail -6; REFACE_HIP_LIB=$L/xsstamp.so RF_ATTN_XS40=$x RF_ATTN_XS80=$x python tools/archive/attn_stamp.py 80 1024 2>&1 | tail -6; done | tee $O/stamps.txt
export REFACE_HIP_LIB=$L/xs.so
for rep in 1 2; do for x in 0 1; do
  echo "XS=$x: $(RF_ATTN_XS40=$x RF_ATTN_XS80=$x python tools/bench_gemm.py --only 'attn d80' --reps 50 2>/dev/null | tail -1)   $(RF_ATTN_XS40=$x RF_ATTN_XS80=$x python tools/bench_gemm.py --only 'attn d40' --reps 30 2>/dev/null | tail -1)"
done; done | tee $O/isolated.txt
bash tools/abenv.sh "RF_ATTN_XS40=0 RF_ATTN_XS80=0" "RF_ATTN_XS40=1 RF_ATTN_XS80=1" "RF_ATTN_XS40=0 RF_ATTN_XS80=0" "RF_ATTN_XS40=1 RF_ATTN_XS80=1" | tee $O/ab_c1.txt
