#!/bin/bash
# same-box A/B of environment settings on the whole bench:  tools/abenv.sh "REFACE_KORDER=0" "REFACE_KORDER=1"
for v in "$@"; do
  env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('env [%s]  %.1f ms/batch  %.3f img/s' % ('$v', r['ms_per_step'], r['value']))"
done
