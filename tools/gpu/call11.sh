#!/bin/bash
set -u
O=gpurun_out/r3c11; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_bf3.py -q -m gpu -p no:cacheprovider -k "obs_conv" > $O/t.txt 2>&1; echo "tests rc=$? $(tail -1 $O/t.txt)"; grep -E "^(FAILED|ERROR)|^E  " $O/t.txt | head
for M in 63 127; do echo "== HAB_BF3=$M"; HAB_BF3=$M timeout 100 python tools/bench_layers.py 1024 2>&1 | grep "conv1 (obs ingest) fwd"; HAB_BF3=$M timeout 100 python tools/bench_layers.py 64 2>&1 | grep "conv1 (obs ingest) fwd"; done
