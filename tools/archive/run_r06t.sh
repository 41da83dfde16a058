#!/bin/bash
# r06t: s_setprio per region in the pipelined attention kernels (prio1: the PV + max region at priority 1, the QK^T + exp region at 0; prio2: the opposite; prio0: none)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06t; O=gpurun_out/r06t; L=$PWD/reface_amd/lib/alt
for rep in 1 2; do for v in prio0 prio1 prio2; do
  echo "$v: $(REFACE_HIP_LIB=$L/$v.so python tools/bench_gemm.py --only 'attn d40' --reps 30 2>/dev/null | tail -1)   $(REFACE_HIP_LIB=$L/$v.so python tools/bench_gemm.py --only 'attn d80' --reps 50 2>/dev/null | tail -1)"
done; done | tee $O/isolated.txt
bash tools/ab_libs.sh $L/prio0.so $L/prio1.so $L/prio2.so $L/prio0.so $L/prio1.so $L/prio2.so | tee $O/ab_c1.txt
