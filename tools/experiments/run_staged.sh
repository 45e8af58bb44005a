#!/bin/bash
# First GPU call of the next round (through gpurun, from the repo root):
#   tools/experiments/run_staged.sh ws      -> the wave-specialised kernel   (bf3_wave_specialised.patch,   HAB_BF3 bit 5)
#   tools/experiments/run_staged.sh pipe2   -> the hand-interleaved schedule (bf3_interleaved_schedule.patch, HAB_BF3_PIPE2)
# Applies the patch to the snapshot on the box (the repo here stays clean), rebuilds the library (~80 s), then
#   1. per-layer timings of the product kernels           -> gpurun_out/staged_<name>_layers_base.txt
#   2. the kernel accuracy / parity tests on the variant  -> gpurun_out/staged_<name>_tests.txt      (under timeout: a barrier
#      mismatch between the two roles would hang the workgroup)
#   3. per-layer timings of the variant                   -> gpurun_out/staged_<name>_layers_*.txt
#   4. whole-cycle bench lines (C2, C3) of the variant    -> gpurun_out/staged_<name>_c{2,3}.json
# Nothing is timed after a failed step 2.  Both experiments in one call: `tools/experiments/run_staged.sh ws; tools/experiments/run_staged.sh pipe2`
# (~25 min of box time together).
set -u
NAME=${1:-ws}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
case $NAME in
  ws)    PATCH=tools/experiments/bf3_wave_specialised.patch;   VARIANTS=("HAB_BF3=63 HAB_TEST_EXTRA_PATH_BITS=32" "HAB_BF3=63 HAB_TEST_EXTRA_PATH_BITS=32 HAB_BF3_WS_TALL=0" "HAB_BF3=63 HAB_TEST_EXTRA_PATH_BITS=32 HAB_BF3_WS_PW=8");;
  pipe2) PATCH=tools/experiments/bf3_interleaved_schedule.patch; VARIANTS=("HAB_BF3_PIPE2=1" "HAB_BF3_PIPE2=2");;
  *) echo "unknown experiment $NAME"; exit 2;;
esac
timeout 120 python tools/bench_layers.py 1024 > $O/staged_${NAME}_layers_base.txt 2>&1
patch -p1 < $PATCH > $O/staged_${NAME}_build.txt 2>&1 || { echo "patch failed"; exit 1; }
make -C habitat-lab_amd/csrc -j8 >> $O/staged_${NAME}_build.txt 2>&1 || { echo "build failed"; tail -20 $O/staged_${NAME}_build.txt; exit 1; }
i=0
for V in "${VARIANTS[@]}"; do
  env $V timeout 600 python -m pytest tests/test_gpu_bf3.py tests/test_gpu_kernels.py -x -q -m gpu > $O/staged_${NAME}_tests_$i.txt 2>&1
  rc=$?
  tail -3 $O/staged_${NAME}_tests_$i.txt
  if [ $rc -ne 0 ]; then echo "variant '$V': tests rc=$rc -- not timed"; i=$((i+1)); continue; fi
  env $V timeout 120 python tools/bench_layers.py 1024 > $O/staged_${NAME}_layers_$i.txt 2>&1
  env $V timeout 300 python bench.py --no-extras > $O/staged_${NAME}_c2_$i.json 2> $O/staged_${NAME}_c2_$i.err
  env $V timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-extras > $O/staged_${NAME}_c3_$i.json 2> $O/staged_${NAME}_c3_$i.err
  if [ $i -eq 0 ]; then  # per-kernel durations + matrix-pipe busy of the first variant (counters in their own pass, under timeout)
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/staged_prof && env $V timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/staged_prof -o layers -- python $R/tools/bench_layers.py 1024 > /dev/null 2>&1)
    DB=$(ls /tmp/staged_prof/*results.db /tmp/staged_prof/*/*results.db 2>/dev/null | head -1)
    [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/staged_${NAME}_kernel_stats.txt > /dev/null
    env $V tools/pmc_run.sh /tmp/staged_pmc python $R/tools/bench_layers.py 256 > /dev/null 2>&1
    python tools/pmc_sq.py /tmp/staged_pmc $O/staged_${NAME}_sq_counters.txt > /dev/null 2>&1
    rm -rf /tmp/staged_prof /tmp/staged_pmc
  fi
  echo "== $V"; paste -d'|' <(cut -c1-70 $O/staged_${NAME}_layers_base.txt) <(cut -c29-70 $O/staged_${NAME}_layers_$i.txt) | head -60
  grep -o '"value": [0-9.]*' $O/staged_${NAME}_c2_$i.json $O/staged_${NAME}_c3_$i.json
  i=$((i+1))
done
# leave the snapshot as it was (a second experiment may follow in the same gpurun call): unapply and rebuild the product library
patch -R -p1 < $PATCH > /dev/null 2>&1
make -C habitat-lab_amd/csrc -j8 > /dev/null 2>&1 || echo "rebuild of the product library failed"
