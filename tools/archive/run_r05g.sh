#!/bin/bash
# r05g: the whole GPU suite as the driver runs it + smoke, on the current library
out=gpurun_out/r05g; mkdir -p $out
timeout 3000 python -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
