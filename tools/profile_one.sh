#!/bin/bash
# rocprofv3 --kernel-trace of ONE workload of bench.py -> gpurun_out/<tag>_<w>_kernel_stats.txt + _phases.txt (development loop; the
# judged artefacts come from tools/collect_profiles.sh).  usage: tools/profile_one.sh <c2|c3|c5> <tag> [steps]
set -u
W=${1:-c3}; TAG=${2:-dev}; STEPS=${3:-3}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_${TAG}_$W
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$W -o $W -- python $R/bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline > $O/prof_${TAG}_$W.log 2>&1
DB=$(ls /tmp/prof_${TAG}_$W/*results.db /tmp/prof_${TAG}_$W/*/*results.db 2>/dev/null | head -1)
python $R/tools/rocprof_summary.py $DB $O/${TAG}_${W}_kernel_stats.txt > /dev/null
python $R/tools/trace_phases.py $DB $O/${TAG}_${W}_phases.txt > /dev/null
grep '^{' $O/prof_${TAG}_$W.log | tail -1 | cut -c1-300
rm -rf /tmp/prof_${TAG}_$W $O/prof_${TAG}_$W.log
