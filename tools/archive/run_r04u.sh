#!/bin/bash
mkdir -p gpurun_out/r04u
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "groupnorm_folded_into_linear" 2>&1 | grep -E "GroupNorm folded|passed|failed|Error" | tee gpurun_out/r04u/pytest_ops.txt
bash tools/abenv.sh "REFACE_GN_FOLD=0" "REFACE_GN_FOLD=1" "REFACE_GN_FOLD=0" "REFACE_GN_FOLD=1" "REFACE_GN_FOLD=0" "REFACE_GN_FOLD=1" 2>&1 | tee gpurun_out/r04u/ab.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-conditioning --no-other-configs --no-parity --profile-json gpurun_out/r04u/prof.json > gpurun_out/r04u/bench.json 2> gpurun_out/r04u/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04u/prof.json'))
for l in d['step_launches']:
    if 'norm.fold' in l['name']: print(f"{l['name'][:50]:50s} {l['ms']*1e3:7.1f} us")
PY
