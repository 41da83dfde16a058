#!/bin/bash
# r04x: (1) the default bench line of the final tree with the final library's PMC passes committed; (2) tile variants of the generic attention kernel at d = 80 / 160
mkdir -p gpurun_out/r04x
python bench.py > gpurun_out/r04x/default_bench.json 2> gpurun_out/r04x/default_bench.log; tail -3 gpurun_out/r04x/default_bench.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04x/default_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["unet_mfma_util_wall"], r["traffic_source"], r["traffic_digest_mismatch"])
PY
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/attn80.so
{
for v in 0 1 2 3 4 5; do echo "RF_ATTN80=$v"; RF_ATTN80=$v python tools/bench_gemm.py --only "attn d80" --reps 50 2>/dev/null; RF_ATTN80=$v python tools/bench_gemm.py --only "attn d80" --cold 1 --reps 30 2>/dev/null; done
for v in 0 1 2 3; do echo "RF_ATTN160=$v"; RF_ATTN160=$v python tools/bench_gemm.py --only "attn d160 N256" --reps 50 2>/dev/null; RF_ATTN160=$v python tools/bench_gemm.py --only "attn d160 N256" --cold 1 --reps 30 2>/dev/null; done
} | tee gpurun_out/r04x/attn_variants.txt
for v in 1 2 3; do RF_ATTN80=$v RF_ATTN160=$v timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention and not x3 and not d40" 2>&1 | tail -1; done | tee gpurun_out/r04x/attn_variants_tests.txt
