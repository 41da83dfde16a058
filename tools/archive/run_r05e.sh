#!/bin/bash
# r05e: tail-round split + restricted patch order: op / bench-shape tests, same-box A/B against the r04 library
out=gpurun_out/r05e; mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "layernorm_folded or bench_shapes_bf16 or linear or geglu" > $out/pytest.log 2>&1; tail -4 $out/pytest.log
tools/ab.sh r04 "" r04 "" > $out/ab.txt 2>&1; cat $out/ab.txt
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof_new.json > $out/bench_new.json 2> $out/bench_new.log
grep "ff.net.0\|GEGLU" -i $out/bench_new.log | head
