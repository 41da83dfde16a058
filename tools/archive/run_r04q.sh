#!/bin/bash
# r04q: final-tree sanity -- smoke(), the default bench line, the whole GPU suite
mkdir -p gpurun_out/r04q
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r04q/smoke.txt
python bench.py > gpurun_out/r04q/default_bench.json 2> gpurun_out/r04q/default_bench.log; tail -4 gpurun_out/r04q/default_bench.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04q/default_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["unet_mfma_util_wall"], r["traffic_source"], r["traffic_digest_mismatch"], d["fusion"], {k:v["value"] for k,v in d["other_configs"].items()})
PY
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r04q/pytest.txt
