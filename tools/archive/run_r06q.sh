#!/bin/bash
# r06q: split-K reduce pass with sixteen loads in flight per lane (slices four at a time) against one slice per iteration: GEMM tests, whole-bench A/B on c1 / c3 / c4
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06q; O=gpurun_out/r06q
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm or conv or linear or split" > $O/pytest_gemm.log 2>&1; tail -2 $O/pytest_gemm.log
B=reface_amd/lib/alt/base_final2.so
bash tools/ab_libs.sh $B - $B - $B - | tee $O/ab_c1.txt
bash tools/ab_libs.sh --config c3 $B - $B - | tee $O/ab_c3.txt
bash tools/ab_libs.sh --config c4 $B - $B - | tee $O/ab_c4.txt
