#!/bin/bash
# c1-only evidence (rocprofv3 stats, bench lines, PMC traffic + matrix-pipe occupancy, default bench line) when the box is one of the pool's fast kind
cd $GRAFT_REPO_ROOT
T=$1; LIMIT=${2:-0}
mkdir -p gpurun_out
ms=$(python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python3 -c "import sys,json; print('%.1f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
echo "probe: $ms ms per batch (limit $LIMIT)" | tee gpurun_out/${T}_probe.txt
if [ "$LIMIT" != "0" ] && python3 -c "import sys; sys.exit(0 if float('$ms') > float('$LIMIT') else 1)"; then echo "slow box: not profiled"; exit 0; fi
EXTRA="" bash tools/profile_round.sh $T c1 2>&1 | tail -4
EXTRA="" bash tools/pmc_mfma.sh ${T}_c1 c1 2>&1 | head -8
python3 bench.py --steps 5 --warmup 2 > gpurun_out/${T}_default_bench.json 2> gpurun_out/${T}_default_bench.log
tail -3 gpurun_out/${T}_default_bench.log
