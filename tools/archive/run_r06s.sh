#!/bin/bash
# r06s: the final library's evidence for configs[3] / configs[4] as for configs[1]: rocprofv3 kernel stats + clean bench line (+ traffic, already in r06final2) and matrix-pipe occupancy
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in c3 c4; do EXTRA="" bash tools/profile_round.sh r06final2 $c 2>&1 | tail -4; bash tools/pmc_mfma.sh r06final2_$c $c 2>&1 | tail -3; done
