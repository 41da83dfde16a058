#!/bin/bash
# r06f: why does the 2-D patch tile order COST the fp8 kernels of configs[4]?  lib/alt/patch8.so = gemm.hip with -DRF_PATCH_FP8=1 (the rule applied
# to the fp8 x fp8 and fp8-weight kernels too); whole batch alternating + the per-launch tables of both libraries on one box (tools/launch_diff.py).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06f; O=gpurun_out/r06f
one() { cfg=$1; tag=$2; shift; shift; env "$@" timeout 900 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag  %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
E=REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/patch8.so
for i in 1 2; do
  one c4 "c4 fp8 in-tree (strip order)  " X=1
  one c4 "c4 fp8 patch order (patch8.so)" $E
done 2>&1 | tee $O/ab_patch_fp8.txt
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs --profile-json $O/c4_strip.json > /dev/null 2>&1
env $E timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs --profile-json $O/c4_patch.json > /dev/null 2>&1
python tools/launch_diff.py $O/c4_strip.json $O/c4_patch.json 1.0 | tee $O/c4_launch_diff.txt
