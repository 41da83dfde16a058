#!/bin/bash
# the fused GEGLU + ff.net.2 kernel (csrc/ffn.hip, C = 320) against the unfused pair, now that the pair's epilogues are known to be store-bound
B="python tools/bench_gemm.py --reps 20 --only"
echo "== cold"; $B "ffn fused" --cold 1 2>&1 | grep -v amdgpu.ids; $B "geglu 320" --cold 1 2>&1 | grep -v amdgpu.ids; $B "ff2+res 1280->320" --cold 1 2>&1 | grep -v amdgpu.ids
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
run REFACE_FFN_FUSE=0
run REFACE_FFN_FUSE=1
run REFACE_FFN_FUSE=0
run REFACE_FFN_FUSE=1
