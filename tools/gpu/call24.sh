#!/bin/bash
set -u
O=gpurun_out/r3c24; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_bf3.py -q -m gpu -p no:cacheprovider -k "wgrad3x3_strip or conv_wgrad_bf3" > $O/t.txt 2>&1; echo "tests rc=$? $(tail -1 $O/t.txt)"; grep -E "^(FAILED|ERROR)|^E  " $O/t.txt | head -20
for V in 0 1; do echo "== HAB_W3B_VARIANT=$V"; HAB_W3B_VARIANT=$V timeout 100 python tools/bench_layers.py 2048 2>&1 | grep -E "wgrad" | grep -E "conv2|conv3|l1"; done
HAB_W3B_VARIANT=1 timeout 300 python -m pytest tests/test_gpu_bf3.py -q -m gpu -p no:cacheprovider -k "wgrad3x3_strip" 2>&1 | tail -1
HAB_BF3=127 timeout 100 python tools/bench_layers.py 2048 2>&1 | grep -E "wgrad" | grep -E "conv2"
