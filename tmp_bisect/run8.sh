#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp tmp_bisect/e9/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip; python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
echo "== e9 adjacent halves, 4 GiB zeroed workspace"; python tmp_bisect/dbg5.py 2>&1 | grep -v amdgpu
echo "== e9 far halves (lane b at +2 GiB)"; LTR_DBG_WB_FAR=1 python tmp_bisect/dbg5.py 2>&1 | grep -v amdgpu
