#!/bin/bash
# round 3: strip-resident 3x3 weight gradient (wgrad3x3_bf3.h): parity, then layer timings with and without bit 7
set -u
O=gpurun_out/r3c17; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_bf3.py -q -m gpu -p no:cacheprovider -k "wgrad3x3_strip or conv_wgrad_bf3" > $O/t.txt 2>&1; echo "tests rc=$? $(tail -1 $O/t.txt)"; grep -E "^(FAILED|ERROR)|^E  " $O/t.txt | head -20
for M in 127 255; do echo "== HAB_BF3=$M"; HAB_BF3=$M timeout 100 python tools/bench_layers.py 2048 2>&1 | grep -E "wgrad" ; done
