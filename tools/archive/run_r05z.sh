#!/bin/bash
# r05z: 3x3 convolutions split by samples at 768x768 (REFACE_SAMPLE_SPLIT): tests, same-box A/B on configs[3] (and configs[1] / [4]: the rule must not fire there)
out=gpurun_out/r05z; mkdir -p $out
timeout 1500 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "sample_split or structural" -s 2>&1 | grep -v Warning | tail -8 | tee $out/pytest.txt
F="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs"
one() { REFACE_SAMPLE_SPLIT=$1 python3 bench.py $F $2 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('REFACE_SAMPLE_SPLIT=$1 %-12s %.1f ms/batch  %.3f img/s  split convs %s  launches %s' % ('$2', r['ms_per_step'], r['value'], r.get('fusion', {}).get('convs_split_by_samples'), r.get('fusion', {}).get('launches_per_ddim_step')))"; }
{ for i in 1 2 3; do one 0 "--config c3"; one 1 "--config c3"; done; one 1 ""; one 1 "--config c4"; } | tee $out/ab.txt
