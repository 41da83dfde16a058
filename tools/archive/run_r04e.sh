#!/bin/bash
# round 4, call E: row-extended A tiles (korder 2) -- op test, full-size bf16 engine tests, whole-bench A/B (REFACE_HX=0 / 1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "row_extended or test_conv" -rA 2>&1 | tail -60 > gpurun_out/r04e/pytest_ops.log
grep -E "passed|failed|korder 2" gpurun_out/r04e/pytest_ops.log | tail -14
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -k "bf16_batch16 or full_width_full_size or c3_engine or fp8c" -rA 2>&1 | tail -40 > gpurun_out/r04e/pytest_full.log
grep -E "passed|failed|rel L2" gpurun_out/r04e/pytest_full.log | tail -12
bash tools/abenv.sh "REFACE_HX=0" "REFACE_HX=1" "REFACE_HX=0" "REFACE_HX=1" > gpurun_out/r04e/ab_hx.log 2>&1
cat gpurun_out/r04e/ab_hx.log
