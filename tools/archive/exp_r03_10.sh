#!/bin/bash
# which launches the ring of four LDS stages helps in situ (exp_r03_9: cfg 6 at M = 1024 + ring on every grid <= 256: -1.3 % per batch)
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
run RF_NOP=1
run RF_GEMM_DEEP=256
run RF_MCFG_M=1024 RF_MCFG_CFG=6 RF_GEMM_DEEP=255
run RF_MCFG_M=1024 RF_MCFG_CFG=6 RF_GEMM_DEEP=256
run RF_NOP=1
run RF_GEMM_DEEP=256
