#!/bin/bash
# same-box A/B of library variants on the whole bench:  tools/ab.sh base "" e1   ("" = the in-tree library)
for v in "$@"; do
  if [ -n "$v" ]; then export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so; else unset REFACE_HIP_LIB; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [%s]  %.1f ms/batch  %.3f img/s' % ('$v', r['ms_per_step'], r['value']))"
done
