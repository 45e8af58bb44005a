#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_bf3.py -q -m gpu -p no:cacheprovider -k "conv2_strip" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-260 | head -12
for M in 255 511; do echo "== HAB_BF3=$M"; HAB_BF3=$M timeout 100 python tools/bench_layers.py 2048 2>&1 | grep -E "conv2" | grep fwd; HAB_BF3=$M timeout 100 python tools/bench_layers.py 64 2>&1 | grep -E "conv2" | grep fwd; done
