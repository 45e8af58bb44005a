#!/bin/bash
for V in 0 1 0 1; do echo -n "c3 TALL64=$V "; HAB_BF3_TALL64=$V timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done
timeout 600 python -m pytest tests/test_gpu_bf3.py tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_policy.py -q -m gpu -p no:cacheprovider -k "golden or fullshape" 2>&1 | tail -2
