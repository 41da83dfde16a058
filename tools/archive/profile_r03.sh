#!/bin/bash
# Round-3 evidence on the GPU box: rocprofv3 stats + bench + PMC traffic + MFMA-busy for c1 / c3 / c4, host scaling probe.
tag=${1:-r03g}
for c in c1 c3 c4; do bash tools/profile_round.sh $tag $c 2>&1 | tail -4; bash tools/pmc_mfma.sh ${tag}_$c $c 2>&1 | head -8; done
python tools/host_scaling_probe.py --procs 8 --batches 6 --device-ms 900 > gpurun_out/${tag}_host_scaling_probe.json 2>/dev/null; cat gpurun_out/${tag}_host_scaling_probe.json
