#!/bin/bash
# tools/profile_final.sh <tag>: the round's final evidence on ONE box for the final library -- GPU suite, smoke, rocprofv3 kernel stats + PMC traffic + matrix-pipe occupancy for configs[1] in bf16
# and fp16, the per-launch profile table -> ceiling budget, the default bench line (with the PMC pass of this library matched by digest).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=${1:-rNNfinal}
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${T}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
EXTRA="" bash tools/profile_round.sh $T c1 2>&1 | tail -6
bash tools/pmc_mfma.sh ${T}_c1 c1 2>&1 | tail -4
EXTRA="" bash tools/profile_round.sh $T c1h 2>&1 | tail -6
bash tools/pmc_mfma.sh ${T}_c1h c1h 2>&1 | tail -4
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs --profile-json gpurun_out/${T}_c1_profile.json > /dev/null 2> gpurun_out/${T}_c1_profile.log
python tools/ceiling_budget.py gpurun_out/${T}_c1_profile.json > gpurun_out/${T}_c1_ceiling_budget.txt; tail -14 gpurun_out/${T}_c1_ceiling_budget.txt
# the committed PMC pass must carry this library's digest for bench.py to report it: copy it where bench.py looks BEFORE the default line is taken
cp gpurun_out/${T}_c1_traffic.json profiles/${T}_c1_pmc_hbm_traffic.json
cp gpurun_out/${T}_c1h_traffic.json profiles/${T}_c1h_pmc_hbm_traffic.json
# ... and the traffic passes of the other configurations the default line reports (configs[3], configs[4] in both fp8 forms)
for c in c3 c4 c4c; do bash tools/pmc_traffic.sh ${T}_$c $c 2>&1 | tail -2; cp gpurun_out/${T}_${c}_traffic.json profiles/${T}_${c}_pmc_hbm_traffic.json; rm -rf gpurun_out/${T}_${c}_pmc_FETCH_SIZE gpurun_out/${T}_${c}_pmc_WRITE_SIZE; done
timeout 1800 python3 bench.py --steps 5 --warmup 2 > gpurun_out/${T}_default_bench.json 2> gpurun_out/${T}_default_bench.log; tail -8 gpurun_out/${T}_default_bench.log
