#!/bin/bash
set -u
O=gpurun_out/r3c27; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_policy.py -q -m gpu -p no:cacheprovider -k "resnet or golden or objectnav or frozen or gaussian" -x > $O/t.txt 2>&1; stamp "tests rc=$? $(tail -1 $O/t.txt)"; grep -E "^(FAILED|ERROR)|^E  " $O/t.txt | head
for V in "HAB_NO_GN_DEFER=1" "HAB_X=0" "HAB_NO_GN_DEFER=1" "HAB_X=0"; do
  env $V timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3.json 2> $O/c3.err; stamp "c3 [$V] $(grep -o '"value": [0-9.]*' $O/c3.json | head -1)"
done
for V in "HAB_NO_GN_DEFER=1" "HAB_X=0"; do
  env $V timeout 300 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5.json 2> $O/c5.err; stamp "c5 [$V] $(grep -o '"value": [0-9.]*' $O/c5.json | head -1)"
done
