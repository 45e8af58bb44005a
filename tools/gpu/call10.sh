#!/bin/bash
for V in "HAB_PL_CFG=0" "HAB_PL_CFG=1" "HAB_PL_CFG=2" "HAB_PL_CFG=1 HAB_PL_ABLATE=4" "HAB_PL_CFG=2 HAB_PL_ABLATE=4"; do
  echo "== $V"
  env $V timeout 120 python tools/bench_layers.py 1024 pl 2>&1 | grep -E "conv2|l2 " | grep -v "fp32 out"
done
