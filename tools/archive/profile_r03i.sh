#!/bin/bash
# final evidence of round 3 after the fused FFN became the default: full GPU suite, then c1 / c3 profile sets (c4 does not run the fused kernel: r03h stands)
python -m pytest tests -q -m gpu 2>&1 | tail -4
for c in c1 c3; do bash tools/profile_round.sh r03i $c 2>&1 | tail -4; bash tools/pmc_mfma.sh r03i_$c $c 2>&1 | head -8; done
