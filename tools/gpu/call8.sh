#!/bin/bash
set -u
O=gpurun_out/r3c8; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for V in "HAB_BWD_SPLIT=0" "HAB_BWD_SPLIT=1" "HAB_BWD_SPLIT=0" "HAB_BWD_SPLIT=1"; do
  env $V timeout 200 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err; stamp "c2 $V $(grep -o '"value": [0-9.]*' $O/c2.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/c2.json | head -1)"
done
timeout 600 python -m pytest tests/test_gpu_policy.py tests/test_gpu_determinism.py -q -m gpu -p no:cacheprovider -x -k "baseline or c1_ or autograd or c2" > $O/policy.txt 2>&1; stamp "tests rc=$? $(tail -1 $O/policy.txt)"
grep -E "^(FAILED|ERROR)|^E  " $O/policy.txt | head -20
