#!/bin/bash
# after the fragment-slab split-K: which neighbouring dispatch rules move now (in situ, same box)
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
run RF_NOP=1
run RF_SHORTK=5200
run RF_SHORTK=3000
run RF_SHORTK=0
run RF_GEMM_DEEP=0
run RF_SK_FRAG=0
run RF_SK256=0
run RF_NOP=1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --profile-json gpurun_out/r03i/prof_c1_frag.json > gpurun_out/r03i/bench_c1_frag.json 2>/dev/null
