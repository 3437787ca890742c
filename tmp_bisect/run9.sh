#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bld() { cp tmp_bisect/$1/* vllm_ltr_amd/csrc/; touch vllm_ltr_amd/csrc/*.hip; python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1; }
bld e10; echo "== e10 (acquire fence at the start of the reduce kernel)"; python tmp_bisect/dbg3.py 2>&1 | grep -v amdgpu | tail -3
bld e11; echo "== e11 (seq_cst fence)"; python tmp_bisect/dbg3.py 2>&1 | grep -v amdgpu | tail -3
bld c; echo "== c, sequence check: which kernels run concurrently (rocprofv3 trace of one n=8 call)"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/r05y_trace -o t -- python $R/tmp_bisect/dbg3.py > /dev/null 2>&1; cd $R
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/r05y_trace/**/*.db", recursive=True)
print(db[:1])
PY
