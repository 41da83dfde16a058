#!/bin/bash
# round 4, GPU call B: whole GPU suite (no -x), sc1-store A/B, configs[4] in the conv-only fp8 mode, host probe with --gpu-prep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
( time timeout 1800 python -m pytest tests -m gpu -q -rA 2>&1 | tail -400 ) > gpurun_out/r04b/pytest.log 2>&1
( timeout 900 bash tools/ab.sh base sc1 base sc1 ) > gpurun_out/r04b/ab_sc1.log 2>&1
( timeout 900 bash tools/abenv.sh "REFACE_LN_FOLD_GEMM=0 REFACE_LN_FOLD=0" "REFACE_LN_FOLD_GEMM=0" "REFACE_LN_FOLD_GEMM=1" "REFACE_LN_FOLD_GEMM=0 REFACE_LN_FOLD=0" "REFACE_LN_FOLD_GEMM=1" ) > gpurun_out/r04b/ab_lnfold.log 2>&1
for dt in fp8c fp8; do
  timeout 600 python bench.py --config c4 --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-conditioning --no-other-configs > gpurun_out/r04b/bench_c4_$dt.json 2> gpurun_out/r04b/bench_c4_$dt.err
done
timeout 400 python tools/host_scaling_probe.py --gpu-prep > gpurun_out/r04b/host_scaling_probe_gpuprep.json 2> gpurun_out/r04b/host_probe.err
timeout 400 python tools/host_scaling_probe.py --gpu-prep --png-level 1 > gpurun_out/r04b/host_scaling_probe_gpuprep_png1.json 2>> gpurun_out/r04b/host_probe.err
grep -E "passed|failed" gpurun_out/r04b/pytest.log | tail -3
cat gpurun_out/r04b/ab_sc1.log gpurun_out/r04b/ab_lnfold.log
cat gpurun_out/r04b/host_scaling_probe_gpuprep*.json
