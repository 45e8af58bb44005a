#!/bin/bash
set -u
O=gpurun_out/r3c26; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for rep in 1 2; do
for V in "HAB_RNN_CHUNKS=4 HAB_FC_DGRAD_OLD=1" "HAB_RNN_CHUNKS=4" "HAB_RNN_CHUNKS=8 HAB_FC_DGRAD_OLD=1" "HAB_RNN_CHUNKS=8" "HAB_RNN_CHUNKS=4 HAB_BF3=127"; do
  env $V timeout 200 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err; stamp "c2 [$V] $(grep -o '"value": [0-9.]*' $O/c2.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/c2.json | head -1)"
done
done
