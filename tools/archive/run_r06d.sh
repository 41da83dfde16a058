#!/bin/bash
# r06d: (1) why fp16 is ~3 % slower on identical kernels (tools/f16_rate_probe.py); (2) tile-rule experiments on the experiment build of gemm.hip
# (REFACE_HIP_LIB = lib/alt/exp.so reads the RF_* switches): the K <= RF_SHORTK rule at 6400 / 7000 for the folded ff.net.2 + proj_out launches (K = 5 C = 6400 at the
# 16x16 level), the 8x8 level on 128x160 tiles; (3) the proj_out fold at configs[3].  Same box, alternating.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06d; O=gpurun_out/r06d
timeout 300 python tools/f16_rate_probe.py > $O/f16_rate_probe.txt 2>&1; cat $O/f16_rate_probe.txt
one() { cfg=$1; tag=$2; shift; shift; env "$@" timeout 900 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag  %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
E=REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
for i in 1 2; do
  one c1 "c1 exp build, defaults          " $E
  one c1 "c1 exp build, RF_SHORTK=6400    " $E RF_SHORTK=6400
  one c1 "c1 exp build, RF_SHORTK=7000    " $E RF_SHORTK=7000
  one c1 "c1 exp build, M=1024 on 128x160 " $E RF_MCFG_M=1024 RF_MCFG_CFG=6
  one c1 "c1 exp build, M=1024 128x160 + SHORTK 6400" $E RF_MCFG_M=1024 RF_MCFG_CFG=6 RF_SHORTK=6400
done 2>&1 | tee $O/ab_tile_rules.txt
for i in 1 2; do
  one c3 "c3 bf16 PO_FOLD=0" REFACE_PO_FOLD=0
  one c3 "c3 bf16 PO_FOLD=1" REFACE_PO_FOLD=1
done 2>&1 | tee $O/ab_po_fold_c3.txt
