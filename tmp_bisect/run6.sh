#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bld() { cp tmp_bisect/$1/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip; LTR_FLAGS_LTR_GEMM="$2" python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; }
bld c ""; echo "== c"; python tmp_bisect/dbg4.py 2>&1 | grep -v amdgpu
echo "== c FORCE_SPLIT=2"; LTR_GEMM_FORCE_SPLIT=2 python tmp_bisect/dbg4.py 2>&1 | grep -v amdgpu
bld c "-DLTR_EPI_STORE=0"; echo "== c plain stores"; python tmp_bisect/dbg4.py 2>&1 | grep -v amdgpu
bld e5 ""; echo "== e5 (reduce kernel: old expression)"; python tmp_bisect/dbg4.py 2>&1 | grep -v amdgpu
bld e6 ""; echo "== e6 (small kernel: old expression)"; python tmp_bisect/dbg4.py 2>&1 | grep -v amdgpu
