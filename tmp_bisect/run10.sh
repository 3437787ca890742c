#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== current tree (reverted = build b)"; python tests/diag/lanes_stress.py 150 2>&1 | grep -v amdgpu | tail -3
cp tmp_bisect/c/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip; python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
echo "== build c (does the stress see it?)"; python tests/diag/lanes_stress.py 60 2>&1 | grep -v amdgpu | tail -3
