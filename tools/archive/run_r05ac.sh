#!/bin/bash
# r05ac: GroupNorm (+ SiLU) applied to the row-extended A tile in LDS (experiment variant gexp against gbase = the tree's gemm.hip, both -DRF_EXPERIMENT)
out=gpurun_out/r05ac; mkdir -p $out
{
REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/gbase.so python3 tools/hx_norm_probe.py 2>&1 | grep -v "Warn\|amdgpu.ids"
for d in 0 1024 3072; do RF_GEMM_DBG=$d REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/gexp.so python3 tools/hx_norm_probe.py 2>&1 | grep -v "Warn\|amdgpu.ids"; done
} | tee $out/probe.txt
F="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs"
one() { RF_GEMM_DBG=$2 REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$1.so python3 bench.py $F 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib $1 RF_GEMM_DBG=$2  %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
{ for i in 1 2; do one gbase 0; one gexp 0; one gexp 1024; one gexp 3072; done; } | tee $out/ab.txt
