#!/bin/bash
mkdir -p gpurun_out/r04r
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r04r/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r04r/pytest.txt
python bench.py --steps 2 --warmup 1 --no-parity --no-other-configs --no-conditioning > gpurun_out/r04r/bench.json 2> gpurun_out/r04r/bench.log; python -c "
import json; d=json.loads(open('gpurun_out/r04r/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['cpu_baseline'])"
