#!/bin/bash
# PMC passes on one microbench case:  tools/pmc_attn.sh "<only-substr>" <tag>
only="$1"; tag=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$n -- python3 tools/bench_gemm.py --only "$only" --reps 3 > gpurun_out/pmc_${tag}_$n.log 2>&1
done
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(lambda: collections.Counter()); cnt=collections.Counter()
for f in glob.glob('gpurun_out/pmc_${tag}_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        if 'rf::' not in k: continue
        tot[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,c in tot.items():
    print(k)
    for n,v in sorted(c.items()): print(f"   {n:28s} {v:16.0f}")
PY
