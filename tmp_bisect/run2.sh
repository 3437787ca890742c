#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in b c; do
  cp tmp_bisect/$v/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip
  python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
  echo "== build $v"; python tmp_bisect/dbg.py 2>&1 | grep -v amdgpu
done
