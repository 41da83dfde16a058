mkdir -p gpurun_out/r03i
python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -3
python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "unet or gemm" 2>&1 | tail -3
bash tools/ab.sh frag "" frag ""
for c in c3; do for v in frag ""; do
  if [ -n "$v" ]; then export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so; else unset REFACE_HIP_LIB; fi
  python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c variant [%s]  %.1f ms/batch  %.3f img/s' % ('$v', r['ms_per_step'], r['value']))"
done; done
export REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/exp.so
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f ms/batch  %.3f img/s' % (r['ms_per_step'], r['value']))"; }
run RF_NOP=1
run RF_MCFG_M=1024 RF_MCFG_CFG=6
run RF_MCFG_M=1024 RF_MCFG_CFG=0
run RF_MCFG_M=1024 RF_MCFG_CFG=3
