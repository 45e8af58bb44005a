#!/bin/bash
# round 3, GPU call 1: new parity tests (odd geometries, determinism) + the wave-specialised kernels (HAB_BF3 bit 5) vs the product kernels
set -u
O=gpurun_out/r3c1; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_policy.py -q -m gpu -k "test_resnet_engine_vs_oracle" -x > $O/odd.txt 2>&1; stamp "odd geometry rc=$? $(tail -1 $O/odd.txt)"
timeout 900 python -m pytest tests/test_gpu_determinism.py -q -m gpu > $O/determinism.txt 2>&1; stamp "determinism rc=$? $(tail -1 $O/determinism.txt)"
timeout 120 python tools/bench_layers.py 1024 > $O/layers_base.txt 2>&1; stamp "layers base"
timeout 200 python bench.py --no-cpu-baseline > $O/c2_base.json 2> $O/c2_base.err; stamp "c2 base $(grep -o '"value": [0-9.]*' $O/c2_base.json | head -1)"
i=0
for V in "HAB_BF3=63 HAB_TEST_EXTRA_PATH_BITS=32" "HAB_BF3=63 HAB_TEST_EXTRA_PATH_BITS=32 HAB_BF3_WS_TALL=0"; do
  env $V timeout 500 python -m pytest tests/test_gpu_bf3.py tests/test_gpu_kernels.py -x -q -m gpu > $O/ws_tests_$i.txt 2>&1
  rc=$?; stamp "variant '$V' tests rc=$rc $(tail -1 $O/ws_tests_$i.txt)"
  if [ $rc -ne 0 ]; then i=$((i+1)); continue; fi
  env $V timeout 120 python tools/bench_layers.py 1024 > $O/layers_ws_$i.txt 2>&1
  env $V timeout 200 python bench.py --no-cpu-baseline > $O/c2_ws_$i.json 2> $O/c2_ws_$i.err; stamp "c2 ws$i $(grep -o '"value": [0-9.]*' $O/c2_ws_$i.json | head -1)"
  if [ $i -eq 0 ]; then
    env $V timeout 200 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3_ws_$i.json 2> $O/c3_ws_$i.err; stamp "c3 ws$i $(grep -o '"value": [0-9.]*' $O/c3_ws_$i.json | head -1)"
  fi
  echo "== $V"; paste -d'|' <(cut -c1-70 $O/layers_base.txt) <(cut -c29-70 $O/layers_ws_$i.txt) | head -70
  i=$((i+1))
done
timeout 200 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3_base.json 2> $O/c3_base.err; stamp "c3 base $(grep -o '"value": [0-9.]*' $O/c3_base.json | head -1)"
