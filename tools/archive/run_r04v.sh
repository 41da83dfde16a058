#!/bin/bash
mkdir -p gpurun_out/r04v
bash tools/abenv.sh "REFACE_GN_FOLD=0" "REFACE_GN_FOLD_MAXC=320" "REFACE_GN_FOLD_MAXC=640" "REFACE_GN_FOLD=0" "REFACE_GN_FOLD_MAXC=320" "REFACE_GN_FOLD_MAXC=640" 2>&1 | tee gpurun_out/r04v/ab.txt
