#!/bin/bash
set -u
O=gpurun_out/r3c29; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
export HAB_NO_GN_DEFER=1 TMPDIR=/tmp
for V in "HAB_BF3=127" "HAB_BF3=255"; do
  env $V timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3.json 2> $O/c3.err; stamp "c3 [$V] $(grep -o '"value": [0-9.]*' $O/c3.json | head -1)"
done
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
DB=$(ls /tmp/kt3/*results.db /tmp/kt3/*/*results.db 2>/dev/null | head -1)
cd $R; python tools/rocprof_summary.py $DB $O/stats.txt > /dev/null; head -16 $O/stats.txt | cut -c1-170
