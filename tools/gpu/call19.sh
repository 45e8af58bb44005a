#!/bin/bash
set -u
O=gpurun_out/r3c19; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_bf3.py -q -m gpu -p no:cacheprovider -k "wgrad3x3_strip" > $O/t.txt 2>&1; echo "tests rc=$? $(tail -1 $O/t.txt)"; grep -E "^(FAILED|ERROR)|^E  " $O/t.txt | head -20
for M in 255; do echo "== HAB_BF3=$M"; HAB_BF3=$M timeout 100 python tools/bench_layers.py 2048 2>&1 | grep -E "wgrad" | grep -E "conv3|l1|l2"; done
