mkdir -p gpurun_out/r03i
python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -5
python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "unet or gemm or vae" 2>&1 | tail -5
bash tools/ab.sh base "" base ""
