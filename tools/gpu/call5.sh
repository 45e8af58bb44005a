#!/bin/bash
set -u
O=gpurun_out/r3c5; mkdir -p $O
for V in "HAB_PL_TAPCM=0" "HAB_PL_TAPCM=1" "HAB_PL_TAPCM=1 HAB_PL_TALL=0" "HAB_PL_TAPCM=1 HAB_PL_TALL=0 HAB_PL_DB=1" "HAB_PL_TAPCM=1 HAB_PL_DB=1"; do
  echo "== $V"; env $V timeout 120 python tools/bench_layers.py 1024 pl 2>&1 | grep "conv2"
done
