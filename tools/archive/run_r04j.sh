#!/bin/bash
# r04j: VAE decode of batch i on a second stream beside the DDIM loop of batch i + 1 (bench.py --overlap-decode), same-box A/B
mkdir -p gpurun_out/r04j
F="--no-cpu-baseline --no-conditioning --no-parity --no-other-configs --no-roofline"
for i in 1 2; do
  for v in base overlap; do
    X=""; [ $v = overlap ] && X="--overlap-decode"
    python bench.py --steps 8 --warmup 2 $F $X > gpurun_out/r04j/${v}_$i.json 2> gpurun_out/r04j/${v}_$i.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04j/${v}_$i.json").read().strip().splitlines()[-1])
print("$v $i", round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms/batch")
PY
  done
done
for cfg in c3 c4; do
  for v in base overlap; do
    X=""; [ $v = overlap ] && X="--overlap-decode"
    python bench.py --config $cfg --steps 5 --warmup 1 $F $X > gpurun_out/r04j/${cfg}_${v}.json 2> gpurun_out/r04j/${cfg}_${v}.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04j/${cfg}_${v}.json").read().strip().splitlines()[-1])
print("$cfg $v", round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms/batch")
PY
  done
done
