#!/bin/bash
set -u
O=gpurun_out/r3c25; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_policy.py -q -m gpu -p no:cacheprovider -k "baseline or golden or lstm_gru or blind" -x > $O/t.txt 2>&1; stamp "tests rc=$? $(tail -1 $O/t.txt)"; grep -E "^(FAILED|ERROR)|^E  " $O/t.txt | head
for V in "HAB_FC_DGRAD_OLD=1" "HAB_X=0" "HAB_FC_DGRAD_OLD=1" "HAB_X=0"; do
  env $V timeout 200 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err; stamp "c2 [$V] $(grep -o '"value": [0-9.]*' $O/c2.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/c2.json | head -1)"
done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c25/c2.json').read().strip().splitlines()[-1])
for r in d['roofline'].get('kernels',[]): print(r['site'], r['ms'], r['calls'], r['tflops'], r['frac'])
PY
