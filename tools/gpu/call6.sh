#!/bin/bash
set -u
for V in "HAB_PL_ABLATE=0" "HAB_PL_ABLATE=1" "HAB_PL_ABLATE=2" "HAB_PL_ABLATE=4" "HAB_PL_ABLATE=5" "HAB_PL_ABLATE=6" "HAB_PL_ABLATE=3"; do
  echo "== $V  (1: no DMA, 2: no MFMA/ds_read, 4: no epilogue)"; env $V timeout 120 python tools/bench_layers.py 1024 pl 2>&1 | grep -E "conv2|conv3|l1 |fc " | grep -v "fp32 out"
done
