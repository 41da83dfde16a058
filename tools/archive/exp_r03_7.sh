export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/pin.so
python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "linear or conv or geglu or groupnorm_stats or split_k" 2>&1 | tail -2
bash tools/ab.sh "" pin "" pin
