#!/bin/bash
# lab: do the two co-resident workgroups of a CU reach their epilogues together, and does de-phasing them at launch pay?
# Builds of ltr_gemm.hip with -DLTR_GEMM_STAGGER=<ticks of 10 ns> -DLTR_GEMM_STAGGER_MASK=<1|32>, diag/gemm_bench at a
# 196,608-token pass (two rounds each), production before and after.   usage: stagger_probe.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-stagger}; mkdir -p $O
B=$R/vllm_ltr_amd/csrc/build/gemm_bench
cd $R
run() { for i in 1 2; do $B 196608 768 3072 10 | grep -v sequence | sed "s/^/$1 round $i: /"; done; }
{
  run production
  for cfg in "2400 32" "2400 1" "1200 32" "3600 32"; do
    set -- $cfg
    LTR_FLAGS_LTR_GEMM="-DLTR_GEMM_STAGGER=$1 -DLTR_GEMM_STAGGER_MASK=$2" python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; run "stagger_$1_mask$2"
  done
  touch vllm_ltr_amd/csrc/ltr_gemm.hip; python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; run production_again
} > $O/stagger.txt 2>&1
cat $O/stagger.txt
