#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in a b c; do
  cp tmp_bisect/$v/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip
  python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
  echo "== build $v"; python -m pytest tests/test_gpu_config1.py -q -m gpu -k "config3 and end_to_end" 2>&1 | tail -2
done
