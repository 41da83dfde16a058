#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
for hx in 0 1; do
  REFACE_HX=$hx python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs --profile-json gpurun_out/r04f/prof_hx$hx.json > gpurun_out/r04f/bench_hx$hx.json 2> gpurun_out/r04f/bench_hx$hx.err
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/r04f/prof_hx0.json'))['step_launches']; b=json.load(open('gpurun_out/r04f/prof_hx1.json'))['step_launches']
print(len(a),len(b))
from collections import defaultdict
g=defaultdict(lambda:[0,0.0,0.0])
for x,y in zip(a,b):
    assert x['name']==y['name']
    if x['family'].startswith('rf_conv_gemm') and x['K']>=2880:
        k=(x['M'],x['N'],x['K']); g[k][0]+=1; g[k][1]+=x['ms']; g[k][2]+=y['ms']
for k,v in sorted(g.items(), key=lambda kv:-kv[1][1]): print(k, v[0], 'hx0 %.1f us  hx1 %.1f us  (%.1f%%)' % (1e3*v[1]/v[0], 1e3*v[2]/v[0], 100*(v[2]/v[1]-1)))
for t in ('hx0','hx1'):
    d=json.loads(open(f'gpurun_out/r04f/bench_{t}.json').read().strip().splitlines()[-1]); print(t, d['value'], d['ms_per_step'], d.get('fusion'))
PY
