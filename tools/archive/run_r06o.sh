#!/bin/bash
# r06o: d = 80 pipelined attention, same-box alternating A/B on configs[3] (768x768: N = 2304 at the d = 80 level) and configs[1]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06o; O=gpurun_out/r06o
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/a80.so
for v in 0 1 0 1; do
  RF_ATTN_PIPE80=$v python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 RF_ATTN_PIPE80=$v  %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"
done | tee $O/ab_c3.txt
bash tools/abenv.sh "RF_ATTN_PIPE80=0" "RF_ATTN_PIPE80=1" "RF_ATTN_PIPE80=0" "RF_ATTN_PIPE80=1" "RF_ATTN_PIPE80=0" "RF_ATTN_PIPE80=1" | tee $O/ab_c1.txt
