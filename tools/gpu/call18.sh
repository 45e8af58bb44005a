#!/bin/bash
# SQ counters of the strip-resident weight gradient (LDS bank conflicts of the transpose reads, MFMA busy)
set -u
O=gpurun_out/r3c18; mkdir -p $O
export HAB_BF3=255
tools/pmc_run.sh /tmp/pmcw3 python $PWD/tools/bench_layers.py 2048 > /dev/null 2>&1
python tools/pmc_sq.py /tmp/pmcw3 $O/sq.txt wgrad3x3_bf3_kernel "ConvWgradProb, 1, 2" > /dev/null
cat $O/sq.txt | head -120
