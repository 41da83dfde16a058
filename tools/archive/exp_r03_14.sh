#!/bin/bash
# bandwidth-bound K = C projections (residual in the epilogue): two co-resident 128x160 blocks per CU instead of one 256x320 block?
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
B="python tools/bench_gemm.py --reps 20 --only"
for e in "RF_NOP=1" "RF_SMALLK_K=640 RF_SMALLK_CFG=6" "RF_SMALLK_K=640 RF_SMALLK_CFG=2" "RF_SMALLK_K=640 RF_SMALLK_CFG=4"; do
  echo "== $e cold"; env $e $B "proj" --cold 1 2>&1 | grep -v amdgpu.ids | grep -v "@8"; env $e $B "qkv 320" --cold 1 2>&1 | grep -v amdgpu.ids
done
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
run RF_NOP=1
run RF_SMALLK_K=640 RF_SMALLK_CFG=6
run RF_SMALLK_K=320 RF_SMALLK_CFG=6
run RF_SMALLK_K=640 RF_SMALLK_CFG=2
run RF_NOP=1
