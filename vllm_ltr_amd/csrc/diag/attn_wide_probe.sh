#!/bin/bash
# lab (round 5): 256-query attention workgroups (attn_f16s_kernel<8, ...>, LTR_ATTN_NW=8) against the 128-query ones (=4) on
# the two BASELINE workloads, alternating on one box; then parity of the wide kernel.   usage: attn_wide_probe.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-attn_wide}; mkdir -p $O
cd $R
COMMON="--no-cpu-baseline --no-unfused --no-strong --no-scale-points --no-class-head --steady-new 0"
line() { python - "$1" <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = o["kernels"]
print(f"{o['value']:9.0f} req/s  {o['ms_per_step']:8.2f} ms  attn {k['attn']['ms_per_step']:7.2f} ms  gemm {k['gemm']['ms_per_step']:8.2f} ms")
PY
}
{
  for nw in 4 8 4 8; do
    LTR_ATTN_NW=$nw python bench.py --steps 5 --warmup 2 $COMMON > $O/b125_$nw.json 2>/dev/null; echo -n "125m sharegpt  NW=$nw: "; line $O/b125_$nw.json
  done
  for nw in 4 8 4 8; do
    LTR_ATTN_NW=$nw python bench.py --model 350m --profile lmsys --steps 2 --warmup 1 $COMMON > $O/b350_$nw.json 2>/dev/null; echo -n "350m lmsys     NW=$nw: "; line $O/b350_$nw.json
  done
  for nw in 4 8; do
    LTR_ATTN_NW=$nw python bench.py --weight-dtype f16-1pass --steps 5 --warmup 2 $COMMON > $O/b1p_$nw.json 2>/dev/null; echo -n "125m one-pass  NW=$nw: "; line $O/b1p_$nw.json
  done
  echo "--- parity with LTR_ATTN_NW=8"
  LTR_ATTN_NW=8 python -m pytest tests/test_gpu_attention.py tests/test_gpu_config1.py tests/test_gpu_outlier.py tests/test_gpu_full_configs.py -x -q -m gpu 2>&1 | tail -3
  LTR_ATTN_NW=8 python -m pytest tests/test_gpu_scorer.py -x -q -m gpu -k "golden or per_layer or chunking or edge or bench_profile" 2>&1 | tail -3
} > $O/attn_wide.txt 2>&1
cat $O/attn_wide.txt
