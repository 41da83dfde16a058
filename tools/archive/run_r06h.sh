#!/bin/bash
# r06h: attn1.to_out in FRONT of the token-resident tail kernel (rf_ffn_desc.wo, REFACE_MID_FUSE): op tests, the full-width UNet tests with it in the launch list, and the
# whole-batch A/B against the rf_conv_gemm launch(es) it replaces; same box, alternating.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06h; O=gpurun_out/r06h
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_pipeline_gpu.py -q -m gpu -k "ffn or attn_in or unet or ddim" > $O/pytest.log 2>&1; tail -6 $O/pytest.log; grep -h "to_out in front" $O/pytest.log | head -8
one() { cfg=$1; tag=$2; shift; shift; env "$@" timeout 900 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag  %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
for i in 1 2 3; do
  one c1 "c1 bf16 MID_FUSE=0" REFACE_MID_FUSE=0
  one c1 "c1 bf16 MID_FUSE=1" REFACE_MID_FUSE=1
done 2>&1 | tee $O/ab_mid_fuse.txt
one c1h "c1h fp16 MID_FUSE=0" REFACE_MID_FUSE=0 | tee -a $O/ab_mid_fuse.txt
one c1h "c1h fp16 MID_FUSE=1" REFACE_MID_FUSE=1 | tee -a $O/ab_mid_fuse.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs --profile-json $O/c1_profile.json > /dev/null 2> $O/c1_profile.log
python - <<'PY'
import json
for r in json.load(open('gpurun_out/r06h/c1_profile.json'))['step_launches']:
    if 'to_out' in r['name'] or 'ff+proj_out' in r['name'] or 'proj_in+norm1' in r['name']:
        print(f"{r['name'][:80]:80s} {r['ms'] * 1e3:8.1f} us")
PY
