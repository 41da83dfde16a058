#!/bin/bash
# r05b: 2-D patch tile order -- op tests, same-box A/B against the r04 library, per-launch fabric reads (FETCH_SIZE) of one c1 DDIM step for both
out=gpurun_out/r05b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "linear or conv or gemm or geglu or layernorm_folded" > $out/pytest_ops.log 2>&1; tail -3 $out/pytest_ops.log
tools/ab.sh r04 "" r04 "" > $out/ab.txt 2>&1; cat $out/ab.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_new -- python3 tools/pmc_per_launch.py --config c1 --list $out/launches_new.json > $out/pmc_new.log 2>&1
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/r04.so
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_r04 -- python3 tools/pmc_per_launch.py --config c1 --list $out/launches_r04.json > $out/pmc_r04.log 2>&1
unset REFACE_HIP_LIB
python3 tools/pmc_per_launch.py --join $out/launches_new.json $out/fetch_new --out $out/per_launch_new.json > $out/per_launch_new.txt 2>&1
python3 tools/pmc_per_launch.py --join $out/launches_r04.json $out/fetch_r04 --out $out/per_launch_r04.json > $out/per_launch_r04.txt 2>&1
tail -2 $out/per_launch_new.txt $out/per_launch_r04.txt
# keep the merge small: the raw counter CSVs hold every dispatch of the model build
find $out -name "*counter_collection.csv" -size +20M -delete
