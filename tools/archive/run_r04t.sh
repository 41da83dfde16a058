#!/bin/bash
# r04t: SpatialTransformer.norm folded into proj_in (rf_groupnorm_fold_linear + per-sample weights): op test, engine parity, same-box A/B
mkdir -p gpurun_out/r04t
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "groupnorm_folded_into_linear or layernorm_folded or linear" 2>&1 | grep -E "GroupNorm folded|passed|failed|Error" | tee gpurun_out/r04t/pytest_ops.txt
timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_pipeline_gpu.py -m gpu -q -s -k "bf16 or c3 or cfg or structural or golden" 2>&1 | grep -E "rel L2|passed|failed|Error|assert" | tee gpurun_out/r04t/pytest_engine.txt
bash tools/abenv.sh "REFACE_GN_FOLD=0" "REFACE_GN_FOLD=1" "REFACE_GN_FOLD=0" "REFACE_GN_FOLD=1" 2>&1 | tee gpurun_out/r04t/ab.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-conditioning --no-other-configs --profile-json gpurun_out/r04t/prof.json > gpurun_out/r04t/bench.json 2> gpurun_out/r04t/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04t/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["fusion"], d["roofline"]["frac"], d["roofline"]["ddim_step_ms_wall"], d.get("parity_bf16_vs_f32",{}).get("psnr_db"), d.get("parity_bf16_vs_f32",{}).get("max_abs"))
PY
