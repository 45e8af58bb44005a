#!/bin/bash
# round 3, GPU call 7: time-major chunked recurrence -- parity, bit-identity with the packed form, C2 cycle time by chunk count
set -u
O=gpurun_out/r3c7; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_planes.py -q -m gpu -p no:cacheprovider -x -k "baseline or c1_ or autograd or lstm_gru" > $O/policy.txt 2>&1; stamp "policy tests rc=$? $(tail -1 $O/policy.txt)"
grep -E "^(FAILED|ERROR)|^E  " $O/policy.txt | head -20
timeout 900 python -m pytest tests/test_gpu_determinism.py -q -m gpu -p no:cacheprovider -k "time_major" > $O/tm.txt 2>&1; stamp "tm identity rc=$? $(tail -1 $O/tm.txt)"
grep -E "^(FAILED|ERROR)|^E  " $O/tm.txt | head -20
for CH in 0 4 8 16; do
  HAB_RNN_CHUNKS=$CH timeout 200 python bench.py --no-cpu-baseline > $O/c2_ch$CH.json 2> $O/c2_ch$CH.err; stamp "c2 chunks=$CH $(grep -o '"value": [0-9.]*' $O/c2_ch$CH.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/c2_ch$CH.json | head -1)"
done
