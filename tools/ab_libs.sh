#!/bin/bash
# same-box alternating A/B of whole-bench runs on library files:  tools/ab_libs.sh [--config cX] <a.so> <b.so> [<a.so> <b.so> ...]   ("-" = the in-tree library)
cfg=""; if [ "$1" = "--config" ]; then cfg="--config $2"; shift 2; fi
for v in "$@"; do
  if [ "$v" = "-" ]; then unset REFACE_HIP_LIB; else export REFACE_HIP_LIB=$(cd $(dirname $v) && pwd)/$(basename $v); fi
  python bench.py $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib [%s] %s  %.1f ms/batch  %.3f img/s' % ('$v', '$cfg', r['ms_per_step'], r['value']))"
done
