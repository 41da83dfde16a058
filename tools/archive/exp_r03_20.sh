#!/bin/bash
# fused FFN at configs[3] (M = 73728 tokens: 576 blocks of 128 = 2.25 rounds of 256 CUs) against the unfused pair
run() { echo "== $*"; env "$@" python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
run REFACE_FFN_FUSE=1
run REFACE_FFN_FUSE=0
run REFACE_FFN_FUSE=1
run REFACE_FFN_FUSE=0
