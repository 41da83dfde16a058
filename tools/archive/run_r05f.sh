#!/bin/bash
# r05f: fused out head + tail-round split: op tests, pipeline tests, same-box A/B (env switches), per-launch profile
out=gpurun_out/r05f; mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gn_silu_conv3x3 or layernorm_folded or geglu" > $out/pytest_ops.log 2>&1; tail -4 $out/pytest_ops.log; grep "fused out head" $out/pytest_ops.log
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -s -k "bench_shapes_bf16 or whole_chain or ddim_vs_reference or unet_full_width or bench_shape" > $out/pytest_pipe.log 2>&1; tail -4 $out/pytest_pipe.log
tools/abenv.sh "REFACE_OUT_FUSE=0" "REFACE_OUT_FUSE=1" "REFACE_OUT_FUSE=0" "REFACE_OUT_FUSE=1" > $out/ab.txt 2>&1; cat $out/ab.txt
tools/ab.sh r04 "" >> $out/ab.txt 2>&1; tail -2 $out/ab.txt
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-conditioning --no-other-configs --profile-json $out/prof_new.json > $out/bench_new.json 2> $out/bench_new.log
grep -i "out\b\|rf_gn_silu\|ff.net.0" $out/bench_new.log | head
