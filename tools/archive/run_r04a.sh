#!/bin/bash
# round 4, GPU call A: full GPU test suite (new gate / resize / x3-attention / fused-LN tests), the store-pattern probe, the default bench
# line (with other_configs) and the fp8 weight-scale ablation
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
( time timeout 1500 python -m pytest tests -m gpu -x -q -rA 2>&1 | tail -120 ) > gpurun_out/r04a/pytest.log 2>&1
timeout 300 python tools/store_probe.py > gpurun_out/r04a/store_probe.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 --profile-json gpurun_out/r04a/prof_c1.json > gpurun_out/r04a/bench_c1.json 2> gpurun_out/r04a/bench_c1.err
timeout 600 python tools/fp8_weight_scale_ablation.py > gpurun_out/r04a/fp8_weight_scale_ablation.json 2> gpurun_out/r04a/fp8_ablation.err
tail -5 gpurun_out/r04a/pytest.log
tail -3 gpurun_out/r04a/bench_c1.err
timeout 400 python tools/host_scaling_probe.py > gpurun_out/r04a/host_scaling_probe.json 2> gpurun_out/r04a/host_probe.err
timeout 400 python tools/host_scaling_probe.py --legacy > gpurun_out/r04a/host_scaling_probe_legacy.json 2>> gpurun_out/r04a/host_probe.err
timeout 400 python tools/host_scaling_probe.py --png-level 1 > gpurun_out/r04a/host_scaling_probe_png1.json 2>> gpurun_out/r04a/host_probe.err
cat gpurun_out/r04a/host_scaling_probe*.json
