#!/bin/bash
# round 3, GPU call 3: pl32 operand planes kernels -- correctness, then layer timings (single / double buffer, tall / short tiles)
set -u
O=gpurun_out/r3c3; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_planes.py -q -m gpu -p no:cacheprovider > $O/planes_tests.txt 2>&1; stamp "planes tests rc=$? $(tail -1 $O/planes_tests.txt)"
grep -E "^(FAILED|ERROR)|^E  " $O/planes_tests.txt | head -40
timeout 300 python -m pytest tests/test_gpu_policy.py -q -m gpu -p no:cacheprovider -k "objectnav_resnet50_256 or 64-128 or refused" > $O/fixed_tests.txt 2>&1; stamp "fixed tests rc=$? $(tail -1 $O/fixed_tests.txt)"
grep -E "^(FAILED|ERROR)|^E  " $O/fixed_tests.txt | head -20
timeout 120 python tools/bench_layers.py 1024 pl > $O/layers_pl.txt 2>&1; stamp "layers pl"; cat $O/layers_pl.txt
HAB_PL_DB=1 timeout 120 python tools/bench_layers.py 1024 pl > $O/layers_pl_db.txt 2>&1; stamp "layers pl db"; cat $O/layers_pl_db.txt
HAB_PL_TALL=0 timeout 120 python tools/bench_layers.py 1024 pl > $O/layers_pl_short.txt 2>&1; stamp "layers pl short"; cat $O/layers_pl_short.txt
HAB_PL_TALL=0 HAB_PL_DB=1 timeout 120 python tools/bench_layers.py 1024 pl > $O/layers_pl_short_db.txt 2>&1; stamp "layers pl short db"; cat $O/layers_pl_short_db.txt
