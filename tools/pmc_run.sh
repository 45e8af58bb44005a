#!/bin/bash
# SQ-counter passes (counters only: no trace domains beside --kernel-trace) of one command.
# usage: tools/pmc_run.sh <outdir> <command...>     then: python tools/pmc_sq.py <outdir>
set -u
OUT=$1; shift
export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_BRANCH"
P3="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
i=0
for PM in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  if [ $i -gt ${PMC_GROUPS:-3} ]; then break; fi
  ( cd /tmp && timeout 300 rocprofv3 --pmc $PM --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- "$@" > $OUT/p$i.log 2>&1 )
  find $OUT/p$i -name "*kernel_trace.csv" -delete
done
