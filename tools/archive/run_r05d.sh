#!/bin/bash
# r05d: in-launch split-K finish (spread over the K slices) -- op tests, same-box A/B (r04 library / new with the reduce pass / new in-launch), exp probe
out=gpurun_out/r05d; mkdir -p $out
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split_k or splitk or conv or linear or gemm" > $out/pytest_ops.log 2>&1; tail -4 $out/pytest_ops.log
python tools/exp_probe.py > $out/exp_probe.txt 2>&1; cat $out/exp_probe.txt
tools/ab.sh r04 "" > $out/ab.txt 2>&1
tools/abenv.sh "REFACE_SK_FIXUP=0" "REFACE_SK_FIXUP=1" "REFACE_SK_FIXUP=0" "REFACE_SK_FIXUP=1" >> $out/ab.txt 2>&1; cat $out/ab.txt
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof_new.json > $out/bench_new.json 2> $out/bench_new.log
REFACE_SK_FIXUP=0 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof_nofix.json > $out/bench_nofix.json 2> $out/bench_nofix.log
