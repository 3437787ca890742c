#!/bin/bash
# lab (round 5): what do the parts of the post-LN fold's RLN epilogue cost on BASELINE config 3?
# (apply diag/rln_probe.patch first: it adds the LTR_RLN_PROBE macros to ltr_gemm.hip)
# Builds of ltr_gemm.hip with -DLTR_RLN_PROBE=<bits> (wrong results): 1 = the residual LayerNorm's gamma / beta as constants (no
# loads in the strip loop), 2 = no statistics combine in the tile prologue (LNC and RLN), 4 = the RLN instances run the ordinary
# one-sweep epilogue (no LayerNorm of the residual at all: the most ANY rewrite of the RLN epilogue could return).
# bench.py --model 350m --profile lmsys, 2 steps, one box, production before and after.   usage: rln_probe.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-rln_probe}; mkdir -p $O
cd $R
COMMON="--model 350m --profile lmsys --steps 2 --warmup 1 --no-cpu-baseline --no-unfused --no-strong --no-scale-points --no-class-head --steady-new 0"
line() { python - "$1" <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = o["kernels"]
print(f"{o['value']:8.0f} req/s  {o['ms_per_step']:8.1f} ms  gemm {k['gemm']['ms_per_step']:8.1f} ms  attn {k['attn']['ms_per_step']:6.1f} ms")
PY
}
{
  python bench.py $COMMON > $O/prod1.json 2>/dev/null; echo -n "production          : "; line $O/prod1.json
  for bits in 1 2 3 4; do
    LTR_FLAGS_LTR_GEMM="-DLTR_RLN_PROBE=$bits" python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
    python bench.py $COMMON > $O/p$bits.json 2>/dev/null; echo -n "LTR_RLN_PROBE=$bits     : "; line $O/p$bits.json
  done
  touch vllm_ltr_amd/csrc/ltr_gemm.hip; python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
  python bench.py $COMMON > $O/prod2.json 2>/dev/null; echo -n "production (again)  : "; line $O/prod2.json
} > $O/rln_probe.txt 2>&1
cat $O/rln_probe.txt
