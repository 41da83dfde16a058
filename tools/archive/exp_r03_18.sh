#!/bin/bash
# ff.net.2 at 64x64 ran 62 us inside the step at r02 and runs 71-80 us in round 3: is it the spread issue of the DMA pieces (RF_SPREAD_DMA 1,
# accepted on the whole bench) that costs the short-K launches?   burst.so = -DRF_SPREAD_DMA=0, exp.so = the default, both experiment builds
B="python tools/bench_gemm.py --reps 20 --only"
for v in exp burst; do
  export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so
  echo "== $v cold"; for c in "ff2" "proj" "geglu" "qkv" "conv3x3 320" "conv3x3 640" "lin big"; do $B "$c" --cold 1 2>&1 | grep -v amdgpu.ids | grep -v "@8"; done
done
bash tools/ab.sh exp burst exp burst
