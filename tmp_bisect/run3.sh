#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp tmp_bisect/c/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip
python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
python tmp_bisect/dbg2.py 2>&1 | grep -v amdgpu
echo "== LTR_STATS_COMB_MIN=1e9"; LTR_STATS_COMB_MIN=1000000000 python tmp_bisect/dbg2.py 2>&1 | grep "repeat\|\[0,8)"
