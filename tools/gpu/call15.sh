#!/bin/bash
# round 3, GPU call 15: full -m gpu suite on the final library of this stage + C2 / C3 bench lines
set -u
O=gpurun_out/r3c15; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_tests.txt 2>&1; stamp "gpu suite rc=$? $(tail -1 $O/gpu_tests.txt)"
grep -E "^(FAILED|ERROR)" $O/gpu_tests.txt | head -20
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
timeout 300 python bench.py --no-extras > $O/c2.json 2> $O/c2.err; stamp "c2 $(grep -o '"value": [0-9.]*' $O/c2.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/c2.json | head -1)"
timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3.json 2> $O/c3.err; stamp "c3 $(grep -o '"value": [0-9.]*' $O/c3.json | head -1)"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c15/c2.json').read().strip().splitlines()[-1])
print({k:d['roofline'][k] for k in ('bound','achieved','frac','avg_launch_ms','share_of_step')})
for r in d['roofline'].get('kernels',[]): print(r['site'], r['ms'], r['tflops'], r['frac'])
print(d.get('parity'))
PY
