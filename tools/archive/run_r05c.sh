#!/bin/bash
# r05c: per-launch fabric reads (FETCH_SIZE) of one c1 DDIM step, patch tile order vs the r04 library, + per-launch times of both on this box
out=gpurun_out/r05c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_new -- python3 tools/pmc_per_launch.py --config c1 --list $out/launches_new.json > $out/pmc_new.log 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof_new.json > $out/bench_new.json 2> $out/bench_new.log
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/r04.so
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_r04 -- python3 tools/pmc_per_launch.py --config c1 --list $out/launches_r04.json > $out/pmc_r04.log 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof_r04.json > $out/bench_r04.json 2> $out/bench_r04.log
unset REFACE_HIP_LIB
python3 tools/pmc_per_launch.py --join $out/launches_new.json $out/fetch_new --out $out/per_launch_new.json > $out/per_launch_new.txt 2>&1
python3 tools/pmc_per_launch.py --join $out/launches_r04.json $out/fetch_r04 --out $out/per_launch_r04.json > $out/per_launch_r04.txt 2>&1
tail -n 2 $out/per_launch_new.txt $out/per_launch_r04.txt
find $out -name "*counter_collection.csv" -size +20M -delete
