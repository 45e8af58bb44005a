#!/bin/bash
for SK in 0 1 2 4 8 16; do
  echo "== skew $SK k-cycles"
  HAB_BF3=7 HAB_BF3_ABLATE=$((SK*256)) timeout 120 python tools/bench_layers.py 1024 2>&1 | grep -E "conv2|fc 25088>512 " | grep -v "algorith"
done
