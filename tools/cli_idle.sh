#!/bin/bash
# GPU-busy fraction of the test-bench CLI with saving ON: kernel trace of a 5-batch synthetic run; the batches
# after the first are steady state.   tools/cli_idle.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cli_idle /tmp/cli_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cli_idle -- python3 scripts/inference_test_bench.py --ckpt none --dataset synthetic --n_items 40 --n_samples 8 --ddim_steps 50 --scale 3.5 --precision bf16 --outdir /tmp/cli_out > gpurun_out/cli_idle.log 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/cli_idle/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# batch boundaries: the first ddim_pack_kernel after a VAE-decode-sized gap in pack kernels -> use to_image kernel (end of a batch's decode)
ends = [i for i, n in enumerate(names) if 'to_image' in n]
print('batches seen:', len(ends))
if len(ends) >= 3:
    a, b = ends[0], ends[-1]                     # steady state: from the end of batch 0 to the end of the last batch
    t0, t1 = int(rows[a]['End_Timestamp']), int(rows[b]['End_Timestamp'])
    busy = 0; cur_s = cur_e = None
    for r in rows[a + 1:b + 1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    n = len(ends) - 1
    print(f'steady state: {n} batches, {(t1 - t0) / n / 1e6:.1f} ms per batch wall, GPU busy {100.0 * busy / (t1 - t0):.1f} % (idle {100.0 - 100.0 * busy / (t1 - t0):.1f} %)')
PY
ls /tmp/cli_out/results | wc -l
rm -rf gpurun_out/cli_idle
