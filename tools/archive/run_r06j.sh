#!/bin/bash
# r06j: the same stage attribution as r06i after the writer went to one job per FILE (long encodes first) with cpus / (world * 4) workers.
# (tools/host_scaling_probe.py --stage reader | writer | both, --device-ms 0: pure host throughput per batch of 8), then the real configuration at today's device time.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06j; O=gpurun_out/r06j
P="python tools/host_scaling_probe.py --procs 8 --batches 6 --natural --gpu-prep --aux-png-level 1"
for st in reader writer both; do $P --device-ms 0 --stage $st > $O/stage_$st.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/stage_$st.json')); print('stage $st  (4 loader workers): 1 process %.0f ms / batch, 8 processes %.0f ms (max)' % (d['1']['ms_per_batch_max'], d['8']['ms_per_batch_max']))"; done
for lw in 6 8; do $P --device-ms 0 --stage reader --loader-workers $lw > $O/stage_reader_lw$lw.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/stage_reader_lw$lw.json')); print('stage reader ($lw loader workers): 1 process %.0f ms / batch, 8 processes %.0f ms (max)' % (d['1']['ms_per_batch_max'], d['8']['ms_per_batch_max']))"; done
for lw in 4 8; do $P --device-ms 805 --loader-workers $lw > $O/real_lw$lw.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/real_lw$lw.json')); print('device 805 ms, $lw loader workers: 1 process %.0f ms / batch, 8 processes %.0f ms (max); %s' % (d['1']['ms_per_batch_max'], d['8']['ms_per_batch_max'], d['verdict']))"; done
