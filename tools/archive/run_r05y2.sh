#!/bin/bash
# r05y (second part, same library digest): after the python-side changes (conv sample split, statistics fusion over slices with different plans) --
# the whole GPU suite, configs[3]'s profile set again (its launch list changed), the default line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05y
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${T}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
EXTRA="" bash tools/profile_round.sh $T c3 2>&1 | tail -4
python3 bench.py --steps 5 --warmup 2 > gpurun_out/${T}_default_bench.json 2> gpurun_out/${T}_default_bench.log
tail -6 gpurun_out/${T}_default_bench.log
