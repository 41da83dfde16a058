#!/bin/bash
# The round's evidence set (tools/archive/profile_r04.sh), taken on a box of the pool's FAST kind when one comes up: boxes differ by ~4-5 % on identical code
# (profiles/r04p_r03_vs_r04_same_box.txt), the slow ones were the ones profiled so far.  $2 = largest ms per batch of the c1 probe to go on with (0 = any box).
cd $GRAFT_REPO_ROOT
T=$1; LIMIT=${2:-0}
mkdir -p gpurun_out
ms=$(python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python3 -c "import sys,json; print('%.1f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
echo "probe: $ms ms per batch (limit $LIMIT)" | tee gpurun_out/${T}_probe.txt
if [ "$LIMIT" != "0" ] && python3 -c "import sys; sys.exit(0 if float('$ms') > float('$LIMIT') else 1)"; then echo "slow box: not profiled"; exit 0; fi
bash tools/archive/profile_r04.sh $T
