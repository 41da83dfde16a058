#!/bin/bash
# r06m: d = 80 pipelined attention after the first-unit overflow fix: tests, PMC counters of the pipelined and the generic kernel on the isolated launch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06m; O=gpurun_out/r06m
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "attention" > $O/pytest_attention.log 2>&1; tail -3 $O/pytest_attention.log
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/a80.so
bash tools/pmc_case.sh "attn d80" a80pipe RF_ATTN_PIPE80=1 > $O/pmc_pipelined.txt 2>&1; cat $O/pmc_pipelined.txt
bash tools/pmc_case.sh "attn d80" a80gen RF_ATTN_PIPE80=0 > $O/pmc_generic.txt 2>&1; cat $O/pmc_generic.txt
