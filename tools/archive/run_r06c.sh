#!/bin/bash
# r06c: the whole GPU suite (fp16 cases included), smoke, the per-launch profile table of c1 (bench.py --profile-json -> tools/ceiling_budget.py), bf16 and fp16
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06c; O=gpurun_out/r06c
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs --profile-json $O/c1_profile.json > $O/c1_bench.json 2> $O/c1_bench.log; tail -3 $O/c1_bench.log
python tools/ceiling_budget.py $O/c1_profile.json > $O/c1_ceiling_budget.txt; tail -3 $O/c1_ceiling_budget.txt
timeout 900 python bench.py --config c1h --steps 3 --warmup 1 --no-cpu-baseline --no-conditioning --no-parity --no-other-configs --profile-json $O/c1h_profile.json > $O/c1h_bench.json 2> $O/c1h_bench.log; tail -3 $O/c1h_bench.log
bash tools/run_r06b.sh
