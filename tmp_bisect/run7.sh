#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bld() { cp tmp_bisect/$1/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip; python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; }
bld e7; echo "== e7 (flag when r_stats_comb is non-null in the reduce kernel)"; python tmp_bisect/dbg3.py 2>&1 | grep -v amdgpu | tail -6
bld e8; echo "== e8 (combine always, overwrite when non-null)"; python tmp_bisect/dbg3.py 2>&1 | grep -v amdgpu | tail -6
