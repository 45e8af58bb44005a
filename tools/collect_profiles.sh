#!/bin/bash
# Collects the measurement artefacts of one round on the GPU box (run through gpurun from the repo root):
#   bench JSON lines (C2 default = the driver's command, C3, C5), rocprofv3 --kernel-trace summaries (C2, C3) and the two
#   --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, counters only) that tools/pmc_traffic.py turns into HBM bytes per launch.
# usage: tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
set -u
TAG=${1:-r01}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/${TAG}_c2_bench.json 2> $O/${TAG}_c2_bench.err
python bench.py --workload c3 --steps 3 --warmup 1 > $O/${TAG}_c3_bench.json 2> $O/${TAG}_c3_bench.err
python bench.py --workload c5 --steps 3 --warmup 1 > $O/${TAG}_c5_bench.json 2> $O/${TAG}_c5_bench.err
cd /tmp
for W in c2 c3; do
  rm -rf $O/prof_${TAG}_$W
  rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_$W -o $W -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_${TAG}_$W.log 2>&1
  python $R/tools/rocprof_summary.py $(ls $O/prof_${TAG}_$W/*results.db $O/prof_${TAG}_$W/*/*results.db 2>/dev/null | head -1) $O/${TAG}_${W}_kernel_stats.txt > /dev/null
  grep '^{' $O/prof_${TAG}_$W.log | tail -1 > $O/${TAG}_${W}_profiled_bench.json
done
rm -rf $O/pmc_${TAG}
for PM in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $PM --kernel-trace -d $O/pmc_${TAG} -o $PM --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python tools/pmc_traffic.py $O/pmc_${TAG} $O/${TAG}_c2_hbm_traffic.json $O/${TAG}_c2_hbm_traffic.txt
rm -rf $O/pmc_${TAG}/*kernel_trace.csv $O/prof_${TAG}_c2 $O/prof_${TAG}_c3
ls -la $O | grep ${TAG}_
