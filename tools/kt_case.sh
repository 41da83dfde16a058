#!/bin/bash
# GPU-side duration of one microbench case (rocprofv3 kernel trace; the event timing of tools/bench_gemm.py has a ~7-12 us host floor).
#   tools/kt_case.sh "<only-substr>" [VAR=value ...]
only="$1"; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_tmp -- python3 tools/bench_gemm.py --only "$only" --reps 10 > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('gpurun_out/kt_tmp/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'rf::' in r['Name']:
            print("   %-86s calls %4s  avg %8.1f us" % (r['Name'].replace('void rf::', '').replace('unsigned short', 'bf16').split('(')[0][:86], r['Calls'], float(r['AverageNs']) / 1e3))
PY
rm -rf gpurun_out/kt_tmp
