#!/bin/bash
# r05h: fused transformer tail (rf_ffn_block): op tests, the rest of the GPU suite from the fp8 quantiser test on, same-box A/B, per-launch profile
out=gpurun_out/r05h; mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -s -k "ffn or quantize_fp8_rows" > $out/pytest_ops.log 2>&1; tail -3 $out/pytest_ops.log; grep "fused tail" $out/pytest_ops.log
tools/abenv.sh "REFACE_TAIL_FUSE=0" "REFACE_TAIL_FUSE=1" "REFACE_TAIL_FUSE=0" "REFACE_TAIL_FUSE=1" > $out/ab.txt 2>&1; cat $out/ab.txt
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof_new.json > $out/bench_new.json 2> $out/bench_new.log
grep -i "rf_ffn\|rf_conv_gemm\|one DDIM" $out/bench_new.log | head
timeout 3000 python -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
