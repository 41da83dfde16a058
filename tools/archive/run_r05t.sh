#!/bin/bash
# r05t: per-launch fabric reads + event-timed durations of one c1 DDIM step on the final library
out=gpurun_out/r05t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/pmc_per_launch.py --config c1 --list $out/launches.json > $out/pmc.log 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof.json > $out/bench.json 2> $out/bench.log
python3 tools/pmc_per_launch.py --join $out/launches.json $out/fetch --out $out/per_launch.json > $out/per_launch.txt 2>&1
tail -n 2 $out/per_launch.txt
find $out -name "*counter_collection.csv" -size +20M -delete
