#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
bash tools/archive/exp_r04_1.sh > gpurun_out/r04d/exp1_afill.log 2>&1
cat gpurun_out/r04d/exp1_afill.log
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "layernorm_folded" -rA 2>&1 | tail -40 > gpurun_out/r04d/pytest_ln.log
tail -5 gpurun_out/r04d/pytest_ln.log
timeout 500 python tools/host_scaling_probe.py --gpu-prep --natural > gpurun_out/r04d/host_probe_natural.json 2> gpurun_out/r04d/host_probe.err
timeout 500 python tools/host_scaling_probe.py --gpu-prep --natural --png-level 1 > gpurun_out/r04d/host_probe_natural_png1.json 2>> gpurun_out/r04d/host_probe.err
cat gpurun_out/r04d/host_probe_natural*.json
