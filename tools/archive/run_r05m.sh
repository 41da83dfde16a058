#!/bin/bash
# r05m: the round's evidence on the FINAL library, one call per part (tools/archive/run_r05m.sh a|b|c)
part=$1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${T:-r05m}
case $part in
a)  # c1: GPU suite, rocprofv3 stats, bench lines, PMC traffic, matrix-pipe occupancy
    timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${T}_pytest_gpu.log
    python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
    EXTRA="" bash tools/profile_round.sh $T c1 2>&1 | tail -4
    EXTRA="" bash tools/pmc_mfma.sh ${T}_c1 c1 2>&1 | head -8
    ;;
b)  # c3 and both c4 forms: bench lines + PMC traffic
    EXTRA="" bash tools/profile_round.sh $T c3 2>&1 | tail -4
    EXTRA="" bash tools/profile_round.sh $T c4 2>&1 | tail -4
    EXTRA="--dtype fp8c" bash tools/pmc_traffic.sh ${T}_c4fp8c c4 | head -6
    rm -rf gpurun_out/${T}_c4fp8c_pmc_FETCH_SIZE gpurun_out/${T}_c4fp8c_pmc_WRITE_SIZE
    ;;
c)  # the default line as the driver runs it (after the PMC passes are committed: traffic matched by digest)
    python3 bench.py --steps 5 --warmup 2 > gpurun_out/${T}_default_bench.json 2> gpurun_out/${T}_default_bench.log
    tail -12 gpurun_out/${T}_default_bench.log
    # the stem kernel against the implicit GEMM + statistics pass, production library, same box
    F="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs"
    for cfg in "" "--config c3" "--config c4"; do for i in 1 2; do for sw in 0 1; do
      REFACE_STEM_FUSE=$sw python3 bench.py $F $cfg 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('REFACE_STEM_FUSE=$sw %-12s %.1f ms/batch  %.3f img/s' % ('$cfg', r['ms_per_step'], r['value']))"
    done; done; done | tee gpurun_out/${T}_stem_ab.txt
    python3 tools/clock_probe.py > gpurun_out/${T}_clock_probe.txt 2>&1; tail -2 gpurun_out/${T}_clock_probe.txt
    # host half of the CLI at 8 processes (no GPU used: the device is a sleep of the measured batch time): the reference's PNG level for every file,
    # results/ at the reference's level + the aux files at level 1 (--fast_aux_png), everything at level 1
    for v in "" "--aux-png-level 1" "--png-level 1"; do
      tagv=$(echo "png6$v" | tr -d ' -')
      python3 tools/host_scaling_probe.py --natural --gpu-prep --device-ms 830 $v > gpurun_out/${T}_host_probe_${tagv}.json 2> gpurun_out/${T}_host_probe_${tagv}.log
      cat gpurun_out/${T}_host_probe_${tagv}.json | cut -c1-400
    done
    ;;
esac
