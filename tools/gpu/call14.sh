#!/bin/bash
for A in 32 45 32; do echo "== HAB_OCP_ABLATE=$A"; HAB_BF3=127 HAB_OCP_ABLATE=$A timeout 100 python tools/bench_layers.py 2048 2>&1 | grep "conv1 (obs ingest) fwd"; done
HAB_BF3=127 HAB_OCP_ABLATE=32 timeout 100 python tools/bench_layers.py 1024 2>&1 | grep "conv1 (obs ingest) fwd"
HAB_BF3=127 HAB_OCP_ABLATE=32 timeout 100 python tools/bench_layers.py 64 2>&1 | grep "conv1 (obs ingest) fwd"
HAB_BF3=127 HAB_OCP_ABLATE=32 timeout 300 python -m pytest tests/test_gpu_bf3.py -q -m gpu -p no:cacheprovider -k "obs_conv_patch" 2>&1 | tail -1
