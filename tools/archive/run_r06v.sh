#!/bin/bash
# r06v: fabric-side reads PER LAUNCH of one configs[1] DDIM step with the final library (the r05t table for round 6's launch list)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06v; O=gpurun_out/r06v
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/pmc_per_launch.py --config c1 --list $O/launches.json > $O/fetch.log 2>&1
python3 tools/pmc_per_launch.py --join $O/launches.json $O/fetch > $O/per_launch_fetch.txt 2>&1; head -40 $O/per_launch_fetch.txt
rm -rf $O/fetch
