#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in x y c; do
  cp tmp_bisect/$v/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip
  python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
  echo "== build $v"; LTR_STATS_COMB_MIN=1000000000 python tmp_bisect/dbg3.py 2>&1 | grep -v amdgpu
done
