#!/bin/bash
mkdir -p gpurun_out/r04ae
python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --no-other-configs > gpurun_out/r04ae/c4_bench.json 2> gpurun_out/r04ae/c4_bench.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04ae/c4_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["dtype"], {k:v for k,v in d.items() if k.startswith("parity")})
PY
