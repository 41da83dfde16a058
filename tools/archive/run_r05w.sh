#!/bin/bash
# r05w: the stem as a pixels-on-lanes kernel (rf_conv3x3_stem): op test, the UNet tests that run through it, same-box A/B against the implicit GEMM + statistics pass
out=gpurun_out/r05w; mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "stem or out_head or gn_silu_conv3x3 or ffn_block" -s 2>&1 | grep -v Warning | tail -15 > $out/pytest_ops.log; tail -8 $out/pytest_ops.log
timeout 2400 python -m pytest tests -x -q -m gpu -k "unet or ddim or engine or cfg" 2>&1 | tail -6 > $out/pytest_unet.log; tail -4 $out/pytest_unet.log
F="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs"
one() { REFACE_STEM_FUSE=$1 python bench.py $F $2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('REFACE_STEM_FUSE=$1 %-12s %.1f ms/batch  %.3f img/s  launches/step %s' % ('$2', r['ms_per_step'], r['value'], r.get('fusion', {}).get('launches_per_ddim_step')))"; }
{ for i in 1 2 3; do one 0 ""; one 1 ""; done; one 0 "--config c3"; one 1 "--config c3"; one 0 "--config c4"; one 1 "--config c4"; } | tee $out/ab.txt
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof.json > $out/bench.json 2> $out/bench.log
python - <<'PY' | tee gpurun_out/r05w/stem_launches.txt
import json
d=json.load(open('gpurun_out/r05w/prof.json'))
for r in d['step_launches']:
    n=r['name']
    if 'input_blocks.0.0' in n or r['family'] in ('rf_groupnorm_stats','rf_conv3x3_stem','rf_gn_silu_conv3x3_small'):
        print(n, r['family'], round(r['ms']*1e3,1),'us')
print({k: (v['calls'], round(v['ms'],4)) for k,v in d['families'].items()})
PY
