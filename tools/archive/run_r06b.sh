#!/bin/bash
# r06b: rf_conv_gemm vs hipBLASLt on the long-K shapes under rocprofv3 -- kernel trace (durations, footprints) and three PMC passes per
# shape (instruction mix, waits, LDS), program directly behind `--`.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06b; O=gpurun_out/r06b
timeout 300 python3 tools/vendor_pmc.py --reps 30 > $O/timing.txt 2>&1; cat $O/timing.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/vendor_pmc.py --reps 10 > $O/trace.log 2>&1
python3 - <<'PY' > gpurun_out/r06b/kernel_trace_summary.txt
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob('gpurun_out/r06b/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not ('rf::' in k or k.startswith('Cijk')): continue
        rows[k].append(r)
for k, rs in rows.items():
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rs)
    r = rs[0]
    print(f"{k[:150]}\n   launches {len(rs)}  median {d[len(d)//2]:.1f} us  min {d[0]:.1f}  grid {r.get('Grid_Size_X','?')} wg {r.get('Workgroup_Size_X','?')} lds {r.get('LDS_Block_Size','?')} "
          f"scratch {r.get('Scratch_Size','?')} vgpr {r.get('VGPR_Count','?')} agpr {r.get('Accum_VGPR_Count','?')} sgpr {r.get('SGPR_Count','?')}")
PY
cat $O/kernel_trace_summary.txt | cut -c1-220
i=0
for shape in 16384x640x5760 4096x1280x11520 65536x320x2880; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${shape}_$i -- python3 tools/vendor_pmc.py --shape $shape --reps 5 > $O/pmc_${shape}_$i.log 2>&1
done
python3 tools/vendor_pmc.py --join $O/pmc_${shape}_* > $O/pmc_${shape}.txt 2>&1; echo "== $shape"; cat $O/pmc_${shape}.txt | cut -c1-200
done
rm -rf $O/pmc_*_[0-9]* $O/trace
