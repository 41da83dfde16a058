mkdir -p gpurun_out/r03i
python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv or linear or split or groupnorm or geglu" 2>&1 | tail -3
python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "unet or gemm" 2>&1 | tail -3
bash tools/ab.sh base "" base ""
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conditioning --profile-json gpurun_out/r03i/prof_c1.json > gpurun_out/r03i/bench_c1.json 2>gpurun_out/r03i/bench_c1.err; tail -c 600 gpurun_out/r03i/bench_c1.json
