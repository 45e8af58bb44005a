#!/bin/bash
# last GPU call of round 3: the C2 line, kernel stats / phases and HBM traffic of the FINAL library
set -u
TAG=r03; R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 280 python bench.py > $O/${TAG}_c2_bench.json 2> $O/${TAG}_c2_bench.err; stamp "c2 $(grep -o '"value": [0-9.]*' $O/${TAG}_c2_bench.json | head -1)"
cd /tmp; rm -rf /tmp/prof_c2
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_c2.log 2>&1
DB=$(ls /tmp/prof_c2/*results.db /tmp/prof_c2/*/*results.db 2>/dev/null | head -1)
python $R/tools/rocprof_summary.py $DB $O/${TAG}_c2_kernel_stats.txt > /dev/null; python $R/tools/trace_phases.py $DB $O/${TAG}_c2_phases.txt > /dev/null
grep '^{' $O/prof_c2.log | tail -1 > $O/${TAG}_c2_profiled_bench.json; rm -f $O/prof_c2.log; stamp trace
rm -rf /tmp/pmc_c2
for PM in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $PM --kernel-trace -d /tmp/pmc_c2 -o $PM --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py /tmp/pmc_c2 $O/${TAG}_c2_hbm_traffic.json $O/${TAG}_c2_hbm_traffic.txt; stamp traffic
