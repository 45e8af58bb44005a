#!/bin/bash
# round 3: full -m gpu suite + C2 / C3 bench lines after the strip-resident weight gradient, VER workers, blind policy
set -u
O=gpurun_out/r3c22; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/gpu_tests.txt 2>&1; stamp "gpu suite rc=$? $(tail -1 $O/gpu_tests.txt)"
grep -E "^(FAILED|ERROR)" $O/gpu_tests.txt | head -20
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
for V in "HAB_BF3=255" "HAB_BF3=127" "HAB_BF3=255"; do
  env $V timeout 200 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err; stamp "c2 [$V] $(grep -o '"value": [0-9.]*' $O/c2.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/c2.json | head -1)"
done
for V in "HAB_BF3=255" "HAB_BF3=127"; do
  env $V timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3.json 2> $O/c3.err; stamp "c3 [$V] $(grep -o '"value": [0-9.]*' $O/c3.json | head -1)"
done
