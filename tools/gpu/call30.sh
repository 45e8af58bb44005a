#!/bin/bash
# round 3, final: full -m gpu suite on the final library, then the C2 artefacts again (bench line, kernel stats, phases, HBM traffic, SQ busy) + C3 / C5 lines
set -u
TAG=r03
R=$PWD
O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/${TAG}_gpu_tests.txt 2>&1; stamp "gpu suite rc=$? $(tail -1 $O/${TAG}_gpu_tests.txt)"
grep -E "^(FAILED|ERROR)" $O/${TAG}_gpu_tests.txt | head -20
cp gpurun_out/parity_margins.json $O/${TAG}_parity_margins.json 2>/dev/null
python bench.py > $O/${TAG}_c2_bench.json 2> $O/${TAG}_c2_bench.err; stamp "c2 $(grep -o '"value": [0-9.]*' $O/${TAG}_c2_bench.json | head -1)"
python bench.py --workload c3 --steps 3 --warmup 1 > $O/${TAG}_c3_bench.json 2> $O/${TAG}_c3_bench.err; stamp "c3 $(grep -o '"value": [0-9.]*' $O/${TAG}_c3_bench.json | head -1)"
python bench.py --workload c5 --steps 3 --warmup 1 > $O/${TAG}_c5_bench.json 2> $O/${TAG}_c5_bench.err; stamp "c5 $(grep -o '"value": [0-9.]*' $O/${TAG}_c5_bench.json | head -1)"
cd /tmp
for W in c2 c3; do
  rm -rf /tmp/prof_${TAG}_$W
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$W -o $W -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_${TAG}_$W.log 2>&1
  DB=$(ls /tmp/prof_${TAG}_$W/*results.db /tmp/prof_${TAG}_$W/*/*results.db 2>/dev/null | head -1)
  python $R/tools/rocprof_summary.py $DB $O/${TAG}_${W}_kernel_stats.txt > /dev/null
  python $R/tools/trace_phases.py $DB $O/${TAG}_${W}_phases.txt > /dev/null
  grep '^{' $O/prof_${TAG}_$W.log | tail -1 > $O/${TAG}_${W}_profiled_bench.json
  rm -rf /tmp/prof_${TAG}_$W $O/prof_${TAG}_$W.log
done
stamp "traces"
rm -rf /tmp/pmc_${TAG}_c2
for PM in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $PM --kernel-trace -d /tmp/pmc_${TAG}_c2 -o $PM --output-format csv -- python $R/bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py /tmp/pmc_${TAG}_c2 $O/${TAG}_c2_hbm_traffic.json $O/${TAG}_c2_hbm_traffic.txt
stamp "traffic"
cd $R
tools/pmc_run.sh /tmp/pmcsq_${TAG} python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline
python tools/pmc_sq.py /tmp/pmcsq_${TAG} $O/${TAG}_c2_sq_counters.txt > /dev/null
stamp "sq"
