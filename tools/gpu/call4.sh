#!/bin/bash
# round 3, GPU call 4: why do the DMA-staged planes kernels plateau where the VALU-staged ones did?  SQ + L2 counters of the layer bench
set -u
R=$PWD; O=$R/gpurun_out/r3c4; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
export TMPDIR=/tmp
tools/pmc_run.sh /tmp/pmc_pl python $R/tools/bench_layers.py 1024 pl > /dev/null 2>&1
python tools/pmc_sq.py /tmp/pmc_pl $O/sq_pl.txt igemm_pl_kernel > /dev/null 2>&1; stamp "sq pl"
for PM in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $PM | cut -d' ' -f1)
  ( cd /tmp && timeout 200 rocprofv3 --pmc $PM --kernel-trace -d /tmp/pmc_tcc/$n -o $n --output-format csv -- python $R/tools/bench_layers.py 1024 pl > /tmp/pmc_tcc_$n.log 2>&1 )
done
python - <<'PY' > $O/tcc_pl.txt 2>&1
import collections, csv, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in glob.glob("/tmp/pmc_tcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\bhab::", "", r["Kernel_Name"])
        if "igemm_pl" not in k and "split_planes" not in k: continue
        a = agg[k[:110]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print("==", k)
    for c, (n, v) in sorted(cs.items()): print(f"   {c:24s} {v / n:16.0f}   ({n} launches)")
PY
stamp "tcc"; cat $O/tcc_pl.txt | head -120
grep -E "^==|MFMA busy|WAIT_ANY /|WAIT_INST_ANY /|ACTIVE_INST_VMEM /|ACTIVE_INST_LDS /|BANK_CONFLICT|INSTS_VALU per|INSTS_MFMA per|INSTS_VMEM_RD per|INSTS_LDS per" $O/sq_pl.txt | head -150
