#!/bin/bash
mkdir -p gpurun_out/r04s
nproc; uptime
timeout 1800 python -m pytest tests -m gpu -q --durations=30 2>&1 | tail -45 | tee gpurun_out/r04s/pytest.txt
uptime
