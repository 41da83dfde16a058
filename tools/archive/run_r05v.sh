#!/bin/bash
# r05v: where the fused feed-forward kernel's time goes -- the kernel alone with pieces removed at compile time (RF_FFN_ABL; results are WRONG, timing only)
out=gpurun_out/r05v; mkdir -p $out
for v in ${VARIANTS:-abl0 abl1 abl2 abl4 abl6 abl8 abl16 abl24 abl31 abl32 abl40 abl0}; do
  for m in ${MS:-65536 32768}; do REFACE_HIP_LIB=$PWD/reface_amd/lib/alt/$v.so python3 tools/ffn_probe.py $m 2>&1 | grep -v "Warning\|amdgpu.ids"; done
done | tee $out/ffn_ablation.txt
