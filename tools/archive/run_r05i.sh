#!/bin/bash
# r05i: three-buffer Wpo stream in the fused tail: op test + A/B against the two-buffer library (alt/tail2.so)
out=gpurun_out/r05i; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "ffn" > $out/pytest_ops.log 2>&1; tail -2 $out/pytest_ops.log
tools/ab.sh tail2 "" tail2 "" > $out/ab.txt 2>&1; cat $out/ab.txt
