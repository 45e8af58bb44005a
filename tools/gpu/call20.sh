#!/bin/bash
set -u
O=gpurun_out/r3c20; mkdir -p $O
export TMPDIR=/tmp HAB_BF3=255
R=$PWD
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/bench_layers.py 2048 > /dev/null 2>&1
DB=$(ls /tmp/kt/*results.db /tmp/kt/*/*results.db 2>/dev/null | head -1)
cd $R; python tools/rocprof_summary.py $DB $O/stats.txt > /dev/null; grep -E "wgrad|colsum|splitk_reduce_kernel<ConvWgrad" $O/stats.txt | cut -c1-200
