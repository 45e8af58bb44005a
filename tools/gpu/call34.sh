#!/bin/bash
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_policy.py tests/test_gpu_fullshape.py -q -m gpu -p no:cacheprovider -k "baseline or golden or blind or lstm_gru or c2" -x 2>&1 | tail -2; stamp tests
for V in 255 511 255 511; do echo -n "c2 HAB_BF3=$V "; HAB_BF3=$V timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done; stamp bench
timeout 600 python -m pytest tests/test_gpu_determinism.py -q -m gpu -p no:cacheprovider -k "c2-2" 2>&1 | tail -2; stamp determinism
