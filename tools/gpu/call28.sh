#!/bin/bash
set -u
O=gpurun_out/r3c28; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
env HAB_NO_GN_DEFER=1 timeout 200 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err; stamp "c2 $(grep -o '"value": [0-9.]*' $O/c2.json | head -1)"
for V in "HAB_NO_GN_DEFER=1" "HAB_X=0"; do
  env $V timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3.json 2> $O/c3.err; stamp "c3 [$V] $(grep -o '"value": [0-9.]*' $O/c3.json | head -1)"
done
