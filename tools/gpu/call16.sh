#!/bin/bash
set -u
O=gpurun_out/r3c16; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for V in "HAB_BF3=127" "HAB_BF3=63" "HAB_BF3=31" "HAB_BF3=127 HAB_RNN_CHUNKS=0" "HAB_BF3=63 HAB_RNN_CHUNKS=0" "HAB_BF3=127"; do
  env $V timeout 200 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err; stamp "c2 [$V] $(grep -o '"value": [0-9.]*' $O/c2.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/c2.json | head -1)"
done
for V in "HAB_BF3=127" "HAB_BF3=31" "HAB_BF3=127"; do
  env $V timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3.json 2> $O/c3.err; stamp "c3 [$V] $(grep -o '"value": [0-9.]*' $O/c3.json | head -1)"
done
