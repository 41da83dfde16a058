#!/bin/bash
# r06a: first GPU pass of round 6 -- the fp16 mode (new kernels instantiations), the proj_out fold, the degree-5 GELU:
# the GPU suite, the activation-range report, the proj_out-fold A/B (same box, alternating), bf16 vs fp16 bench lines.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06a; O=gpurun_out/r06a
timeout 3300 python -m pytest tests/ -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python tools/act_range.py > $O/act_range.txt 2> $O/act_range.err; head -4 $O/act_range.txt; grep '^##' $O/act_range.txt
one() { timeout 900 python bench.py --config $1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2  %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
for i in 1 2 3; do
  REFACE_PO_FOLD=0 one c1 "c1 bf16 PO_FOLD=0"
  REFACE_PO_FOLD=1 one c1 "c1 bf16 PO_FOLD=1"
done 2>&1 | tee $O/ab_po_fold.txt
REFACE_PO_FOLD=1 one c1h "c1h fp16 PO_FOLD=1" | tee -a $O/ab_po_fold.txt
timeout 1500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-conditioning > $O/default_bench.json 2> $O/default_bench.log; tail -12 $O/default_bench.log
