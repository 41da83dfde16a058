#!/bin/bash
# r04y: d = 80 attention on 128-key stages / 8 waves -- op tests, same-box A/B against the previous library (reface_amd/lib/alt/base.so), c1 and c3
mkdir -p gpurun_out/r04y
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "attention" 2>&1 | tail -2 | tee gpurun_out/r04y/pytest.txt
B=$PWD/reface_amd/lib/alt/base.so
bash tools/abenv.sh "REFACE_HIP_LIB=$B" "A=new" "REFACE_HIP_LIB=$B" "A=new" 2>&1 | tee gpurun_out/r04y/ab_c1.txt
for v in "REFACE_HIP_LIB=$B" "A=new" "REFACE_HIP_LIB=$B" "A=new"; do
  env $v python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 env [%s]  %.1f ms/batch  %.3f img/s' % ('$v'[-12:], r['ms_per_step'], r['value']))"
done | tee gpurun_out/r04y/ab_c3.txt
