#!/bin/bash
# (no TA_* / TCP_* sets: that pass aborts inside rocprofv3 on this pool and hangs until the call is killed)
# PMC passes (one rocprofv3 run per counter set) on one microbench case:  tools/pmc_case.sh "<only-substr>" <tag> [VAR=value ...]
only="$1"; tag=$2; shift; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 tools/bench_gemm.py --only "$only" --reps 3 > gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(lambda: collections.Counter()); n=collections.defaultdict(lambda: collections.Counter())
for f in glob.glob('gpurun_out/pmc_${tag}_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:70]
        if 'rf::' not in k: continue
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
for k,c in tot.items():
    print(k)
    for name,v in sorted(c.items()): print(f"   {name:36s} {v / n[k][name]:16.0f}   per launch ({n[k][name]} launches)")
PY
rm -rf gpurun_out/pmc_${tag}_[0-9]*
