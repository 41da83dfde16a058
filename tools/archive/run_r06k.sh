#!/bin/bash
# r06k: the host half once more with what r06i / r06j found -- the box's container runs under a CFS quota of 16 CPUs (cpu.max 1600000 100000; the affinity mask says 256), so
# "8 processes at once" on this box is 8 ranks on ONE GPU slot's CPU share.  Worker pools are sized by the quota now (output.available_cpus).  Reader stage over 24 batches (the
# 6-batch runs of r06i / r06j finish inside the DataLoader's prefetch depth and measure nothing), writer / both stages, and one rank alone at today's device time.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06k; O=gpurun_out/r06k
python tools/host_cpu_probe.py > $O/host_cpu_probe.txt 2>&1; cat $O/host_cpu_probe.txt
P="python tools/host_scaling_probe.py --natural --gpu-prep --aux-png-level 1"
$P --procs 2 --batches 24 --device-ms 0 --stage reader > $O/stage_reader_24.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/stage_reader_24.json')); print('stage reader, 24 batches: 1 process %.0f ms / batch, 2 processes %.0f ms (max); quota %d cpus' % (d['1']['ms_per_batch_max'], d['2']['ms_per_batch_max'], d['cpus_under_cgroup_quota']))"
for n in 2 4 8; do $P --procs $n --batches 8 --device-ms 0 --stage writer > $O/stage_writer_p$n.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/stage_writer_p$n.json')); print('stage writer: 1 process %.0f ms / batch, $n processes %.0f ms (max)' % (d['1']['ms_per_batch_max'], d['$n']['ms_per_batch_max']))"; done
for n in 2 4 8; do $P --procs $n --batches 8 --device-ms 805 > $O/real_p$n.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/real_p$n.json')); print('device 805 ms: 1 process %.0f ms / batch, $n processes %.0f ms (max)' % (d['1']['ms_per_batch_max'], d['$n']['ms_per_batch_max']))"; done
