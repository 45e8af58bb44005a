#!/bin/bash
# Collects the measurement artefacts of one round on the GPU box (run through gpurun from the repo root):
#   * bench JSON lines: C2 default (= the driver's command), C3, C5
#   * rocprofv3 --kernel-trace summaries (C2, C3, C5) + the rollout / learner phase split of the same traces (tools/trace_phases.py)
#   * --pmc passes, counters only, each in its own run (MI355X_MICROARCH.md):
#       FETCH_SIZE, WRITE_SIZE (C2 and C3)            -> tools/pmc_traffic.py -> HBM bytes per launch
#       SQ busy / MFMA-busy / wait / instruction mix  -> tools/pmc_sq.py      -> per-kernel matrix-pipe utilisation (C2: all three
#                                                        counter groups; C3, C5: the busy / MFMA-busy group)
#   * the CPU leg of bench.py at 32 and 64 host threads as well (the default line uses 16) -> <tag>_cpu_leg_threads.json
# Every profiler pass runs under `timeout`: a rocprofv3 --pmc pass of the C3 command once hung until gpurun's limit.
# usage: tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
set -u
TAG=${1:-r06}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_c2_bench.json 2> $O/${TAG}_c2_bench.err   # the driver's command
python bench.py --workload c3 --steps 10 --warmup 2 > $O/${TAG}_c3_bench.json 2> $O/${TAG}_c3_bench.err
python bench.py --workload c5 --steps 5 --warmup 2 > $O/${TAG}_c5_bench.json 2> $O/${TAG}_c5_bench.err
cd /tmp
for W in c2 c3 c5; do
  rm -rf /tmp/prof_${TAG}_$W
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$W -o $W -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_${TAG}_$W.log 2>&1
  DB=$(ls /tmp/prof_${TAG}_$W/*results.db /tmp/prof_${TAG}_$W/*/*results.db 2>/dev/null | head -1)
  python $R/tools/rocprof_summary.py $DB $O/${TAG}_${W}_kernel_stats.txt > /dev/null
  python $R/tools/trace_phases.py $DB $O/${TAG}_${W}_phases.txt > /dev/null
  grep '^{' $O/prof_${TAG}_$W.log | tail -1 > $O/${TAG}_${W}_profiled_bench.json
  rm -rf /tmp/prof_${TAG}_$W $O/prof_${TAG}_$W.log
done
# (C5: `rocprofv3 --pmc` of the ObjectNav command hung in three of four attempts in round 5 -- FETCH_SIZE, WRITE_SIZE and the SQ group each
#  ran into their `timeout` on different boxes -- and cost 10 GPU-minutes per collection; its counter passes are run only with PMC_C5=1)
for W in c2 c3 ${PMC_C5:+c5}; do
  rm -rf /tmp/pmc_${TAG}_$W
  for PM in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $PM --kernel-trace -d /tmp/pmc_${TAG}_$W -o $PM --output-format csv -- python $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  done
  python $R/tools/pmc_traffic.py /tmp/pmc_${TAG}_$W $O/${TAG}_${W}_hbm_traffic.json $O/${TAG}_${W}_hbm_traffic.txt
  rm -rf /tmp/pmc_${TAG}_$W
done
cd $R
tools/pmc_run.sh /tmp/pmcsq_${TAG} python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline
python tools/pmc_sq.py /tmp/pmcsq_${TAG} $O/${TAG}_c2_sq_counters.txt > /dev/null
rm -rf /tmp/pmcsq_${TAG}
for W in c3 ${PMC_C5:+c5}; do
  PMC_GROUPS=1 tools/pmc_run.sh /tmp/pmcsq_${TAG}_$W python $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline
  python tools/pmc_sq.py /tmp/pmcsq_${TAG}_$W $O/${TAG}_${W}_sq_counters.txt > /dev/null
  rm -rf /tmp/pmcsq_${TAG}_$W
done
# the CPU leg at further thread settings (the default line uses 16; every core of the 256-thread host did not finish in 400 s in round 3)
for TH in 32 64; do
  timeout 300 python bench.py --steps 1 --warmup 1 --no-extras --cpu-threads $TH > $O/${TAG}_c2_cpu_${TH}.json 2> $O/${TAG}_c2_cpu_${TH}.err
done
python - <<PY
import json
rows = []
for f in ("$O/${TAG}_c2_bench.json", "$O/${TAG}_c2_cpu_32.json", "$O/${TAG}_c2_cpu_64.json"):
    try:
        c = json.loads(open(f).read().strip().splitlines()[-1])["cpu_baseline"]
        rows.append({"threads": c["cores"], "env_steps_per_s": c["value"]})
    except Exception as e:
        print("cpu leg missing in", f, e)
rows.append({"threads": 256, "env_steps_per_s": None, "note": "not finished after 400 s (round 3 measurement, < 20.5 env-steps/s)"})
json.dump(rows, open("$O/${TAG}_cpu_leg_threads.json", "w"))
print(rows)
PY
ls -la $O | grep ${TAG}_
