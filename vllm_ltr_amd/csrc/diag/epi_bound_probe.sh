#!/bin/bash
# VERDICT r4 item 3, first the bound: what can an accumulator-freeing epilogue return on the K = 768 GEMM shapes?
# Three builds of ltr_gemm.hip on the GPU box, the four dense layers of a decoder layer at a 196,608-token pass through
# diag/gemm_bench (each shape 10 launches back to back; two alternating rounds per build):
#   production        | -DLTR_GEMM_NOSTORE : epilogue without its split-plane stores (QKV, fc1)
#   -DLTR_GEMM_EPI_1IN8=8 : only 1 tile in 8 of QKV / fc1 runs any epilogue at all (the K loops alone)
# usage: epi_bound_probe.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-epi_bound}; mkdir -p $O
B=$R/vllm_ltr_amd/csrc/build/gemm_bench
cd $R
run() { for i in 1 2; do $B 196608 768 3072 10 | sed "s/^/$1 round $i: /"; done; }
{
  run production
  LTR_FLAGS_LTR_GEMM="-DLTR_GEMM_NOSTORE" python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; run nostore
  LTR_FLAGS_LTR_GEMM="-DLTR_GEMM_EPI_1IN8=8" python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; run epi_1in8
  LTR_FLAGS_LTR_GEMM="-DLTR_GEMM_EPI_1IN8=1000000" python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; run epi_none
  touch vllm_ltr_amd/csrc/ltr_gemm.hip; python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; run production_again
} > $O/epi_bound.txt 2>&1
cat $O/epi_bound.txt
