#!/bin/bash
# in-situ A/B (whole c1 bench, same box) of the tile choice at the 8x8 level (M = 1024): cold weights (29.5-59 MB per conv, streamed once)
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
run RF_NOP=1
run RF_MCFG_M=1024 RF_MCFG_CFG=6
run RF_MCFG_M=1024 RF_MCFG_CFG=6 RF_GEMM_DEEP=256
run RF_MCFG_M=1024 RF_MCFG_CFG=0
run RF_NOP=1
