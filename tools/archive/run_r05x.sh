#!/bin/bash
# r05x: rf_conv3x3_stem variants alone (tools/stem_probe.py): stem0 = first version (weight table by a load -> LDS-write loop), stem1 = every weight load in flight first,
# stem2 = stem1 + the output tile through LDS (whole rows leave as runs of 16-byte vectors)
out=gpurun_out/r05x; mkdir -p $out
for v in ${VARIANTS:-stem0 stem1 stem2}; do
  REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so python3 tools/stem_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids"
  B=4 HW=96 REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so python3 tools/stem_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids"
done | tee $out/probe.txt
for v in ${TESTV:-stem1 stem2}; do
  REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "stem" 2>&1 | tail -2
done | tee $out/pytest.txt
