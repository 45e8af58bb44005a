#!/bin/bash
# Where does the time of the split-bf16 contraction kernels go?  Ablation (cdna_hip_programming.md 5.4 rule 8): the same launches with
# parts of the kernel removed by a block-uniform flag (HAB_BF3_ABLATE for igemm_bf3_kernel, HAB_PL_ABLATE for igemm_pl_kernel):
#   1 no operand gathers / DMA   2 no fragment reads + MFMAs   4 no epilogue (no output traffic)   8 no split + LDS writes (bf3 only)
# usage (GPU box): tools/ablate_layers.sh [frames] > profiles/r03_ablation.txt
B=${1:-1024}
export HAB_BF3=31  # plain kernels everywhere (no patch / ws variants): the ablation flags live in igemm_bf3_kernel
for A in 0 1 2 4 8 9 6 13 11; do
  echo "== igemm_bf3_kernel HAB_BF3_ABLATE=$A (HAB_BF3=7: im2col form for every layer)"
  HAB_BF3=7 HAB_BF3_ABLATE=$A timeout 120 python tools/bench_layers.py $B 2>&1 | grep -E "conv2|conv3|l1 |fc 25088>512 " | grep -v "algorith"
done
