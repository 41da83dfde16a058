#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
L=$PWD/reface_amd/lib/alt/base.so
run() { tag=$1; shift; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs --profile-json gpurun_out/r04g/prof_$tag.json > gpurun_out/r04g/bench_$tag.json 2> gpurun_out/r04g/bench_$tag.err; }
run base REFACE_HIP_LIB=$L REFACE_HX=0 RF_GEMM_DBG=0
run askip REFACE_HIP_LIB=$L REFACE_HX=0 RF_GEMM_DBG=256
run hx REFACE_HIP_LIB=$L REFACE_HX=1 RF_GEMM_DBG=0
run hxnoa REFACE_HIP_LIB=$L REFACE_HX=1 RF_GEMM_DBG=512
python - <<'PY'
import json
from collections import defaultdict
P={t:json.load(open(f'gpurun_out/r04g/prof_{t}.json'))['step_launches'] for t in ('base','askip','hx','hxnoa')}
g=defaultdict(lambda:[0,{}])
for i,x in enumerate(P['base']):
    if x['family'].startswith('rf_conv_gemm') and x['K']>=2880:
        k=(x['M'],x['N'],x['K']); g[k][0]+=1
        for t in P: g[k][1][t]=g[k][1].get(t,0.0)+P[t][i]['ms']
for k,v in sorted(g.items(), key=lambda kv:-kv[1][1]['base']):
    n=v[0]; print(k,n,'  '.join('%s %.1f' % (t,1e3*v[1][t]/n) for t in ('base','askip','hx','hxnoa')))
for t in P:
    d=json.loads(open(f'gpurun_out/r04g/bench_{t}.json').read().strip().splitlines()[-1]); print(t, round(d['value'],3), round(d['ms_per_step'],1), d.get('fusion',{}).get('convs_on_row_extended_a_tiles'))
PY
