#!/bin/bash
# r04o: split-K finished inside the GEMM launch -- op tests, then same-box A/B of the whole bench (REFACE_SK_FIXUP=0 keeps the reduce passes)
mkdir -p gpurun_out/r04o
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "split_k or splitk or stats or conv" 2>&1 | tail -6 | tee gpurun_out/r04o/pytest_ops.txt
bash tools/abenv.sh "REFACE_SK_FIXUP=0" "REFACE_SK_FIXUP=1" "REFACE_SK_FIXUP=0" "REFACE_SK_FIXUP=1" 2>&1 | tee gpurun_out/r04o/ab.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs > gpurun_out/r04o/bench.json 2> gpurun_out/r04o/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04o/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["fusion"], d["roofline"]["frac"], d["roofline"]["ddim_step_ms_wall"])
PY
