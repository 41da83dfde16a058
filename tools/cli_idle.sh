#!/bin/bash
# Steady-state batch time of the test-bench CLI with saving ON, from its own per-batch host timing
# (RF_CLI_TIMING=1: loader / enqueue / flush per batch; their sum is the wall time of a batch, since the enqueue phase blocks on the
# previous batch at its first host -> device copy).  Compare with bench.py's ms_per_step (sampling + decode of the same batch) plus its
# `conditioning_bf16.ms_per_batch`: the difference is what the GPU idles per batch.  (A rocprofv3 kernel trace of the CLI inflates the
# host side -- 1.47 s per batch, "30 % idle" -- and is not usable for this.)      tools/cli_idle.sh
cd $GRAFT_REPO_ROOT
rm -rf /tmp/cli_out
RF_CLI_TIMING=1 python3 scripts/inference_test_bench.py --config configs/reface_inference.yaml --ckpt none --dataset synthetic --n_items 64 --n_samples 8 \
    --ddim_steps 50 --scale 3.5 --precision bf16 --outdir /tmp/cli_out 2>&1 | grep "\[cli\]" | python3 -c "
import re, sys
tot = [sum(float(x) for x in re.findall(r'(\d+) ms', l)) for l in sys.stdin]
ss = tot[2:]
print(f'batches {len(tot)}; steady state (from the 3rd): {sum(ss) / len(ss):.0f} ms per batch of 8 = {8e3 * len(ss) / sum(ss):.2f} images/s with PNG saving on')"
ls /tmp/cli_out/results | wc -l
