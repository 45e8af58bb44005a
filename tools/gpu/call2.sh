#!/bin/bash
# round 3, GPU call 2: the full -m gpu suite with the tightened tolerances and the new cases
set -u
O=gpurun_out/r3c2; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_tests.txt 2>&1; echo "gpu suite rc=$? $(tail -1 $O/gpu_tests.txt)"
grep -E "^(FAILED|ERROR)" $O/gpu_tests.txt | head -40
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
cp gpurun_out/parity_c5.json $O/ 2>/dev/null
