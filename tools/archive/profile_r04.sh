#!/bin/bash
# round 4 evidence on the GPU box: full GPU suite, then per config rocprofv3 kernel stats + bench lines + PMC traffic + PMC matrix-pipe occupancy,
# and the same for the parity mode's fast form (c1 --dtype f32x3)
cd $GRAFT_REPO_ROOT
T=${1:-r04c}
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu -rA 2>&1 | tail -300 ) > gpurun_out/${T}_pytest.log 2>&1
grep -E " passed| failed" gpurun_out/${T}_pytest.log | tail -2
for c in c1 c3 c4; do EXTRA="" bash tools/profile_round.sh $T $c 2>&1 | tail -4; EXTRA="" bash tools/pmc_mfma.sh ${T}_$c $c 2>&1 | head -8; done
# the parity mode's fast form: kernel stats, per-family table, matrix-pipe occupancy
export EXTRA="--dtype f32x3"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_c1_f32x3_rocprof -- python3 bench.py --config c1 --steps 1 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs $EXTRA > gpurun_out/${T}_c1_f32x3_bench_under_rocprof.json 2> gpurun_out/${T}_c1_f32x3_rocprof.log
cp $(ls gpurun_out/${T}_c1_f32x3_rocprof/*/*kernel_stats.csv | head -1) gpurun_out/${T}_c1_f32x3_rocprofv3_kernel_stats.csv
rm -rf gpurun_out/${T}_c1_f32x3_rocprof
python3 bench.py --config c1 --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --no-other-configs $EXTRA --profile-json gpurun_out/${T}_c1_f32x3_prof.json > gpurun_out/${T}_c1_f32x3_bench.json 2> gpurun_out/${T}_c1_f32x3_bench.log
bash tools/pmc_mfma.sh ${T}_c1_f32x3 c1 2>&1 | head -8
export EXTRA=""
# the default line the driver will run (other_configs included)
python3 bench.py --steps 5 --warmup 2 --profile-json gpurun_out/${T}_c1_prof.json > gpurun_out/${T}_default_bench.json 2> gpurun_out/${T}_default_bench.log
tail -4 gpurun_out/${T}_default_bench.log
# shader clock actually held inside the DDIM loop (tools/clock_probe.*): evidence for the power-management reading of the stale-operand experiments
timeout 300 python3 tools/clock_probe.py > gpurun_out/${T}_clock_probe.txt 2>&1
tail -2 gpurun_out/${T}_clock_probe.txt
