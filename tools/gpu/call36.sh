#!/bin/bash
timeout 200 python -m pytest tests/test_gpu_bf3.py -q -m gpu -p no:cacheprovider -k "conv2_strip" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-260 | head -8
for M in 511 1023; do echo "== HAB_BF3=$M"; HAB_BF3=$M timeout 60 python tools/bench_layers.py 512 2>&1 | grep -E "conv2" | grep dgrad; done
for V in 511 1023 511 1023; do echo -n "c2 HAB_BF3=$V "; HAB_BF3=$V timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done
