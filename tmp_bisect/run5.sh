#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp tmp_bisect/c/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip
python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
for e in "X=1" "LTR_NO_LN_FOLD=1" "LTR_GEMM_SMALL_M=0" "LTR_GEMM_FORCE_SPLIT=1" "LTR_NO_LASTQ=1" "LTR_ATTN_SPLITKV_TOKENS=0" "LTR_LANE_PROBE=0" "AMD_SERIALIZE_KERNEL=3" "HIP_LAUNCH_BLOCKING=1"; do
  echo "== $e"; env $e python tmp_bisect/dbg3.py 2>&1 | grep -v amdgpu
done
