#!/bin/bash
# r06e: attention.hip compiled with -fno-honor-nans (lib/alt/attn_nonan.so: the online-softmax max chains lose their NaN-quieting v_max x, x -- 10 of ~86 VALU
# instructions per 64 x 32 unit of the d = 40 kernel): the attention tests on that library, then the whole-batch A/B at configs[1] and configs[3], same box, alternating.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06e; O=gpurun_out/r06e
REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/attn_nonan.so timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "attention" > $O/pytest_attention_nonan.log 2>&1; tail -3 $O/pytest_attention_nonan.log
one() { cfg=$1; tag=$2; shift; shift; env "$@" timeout 900 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag  %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
E=REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/attn_nonan.so
for i in 1 2 3; do
  one c1 "c1 in-tree library          " X=1
  one c1 "c1 attention -fno-honor-nans" $E
done 2>&1 | tee $O/ab_attn_nonan.txt
for i in 1 2; do
  one c3 "c3 in-tree library          " X=1
  one c3 "c3 attention -fno-honor-nans" $E
done 2>&1 | tee -a $O/ab_attn_nonan.txt
