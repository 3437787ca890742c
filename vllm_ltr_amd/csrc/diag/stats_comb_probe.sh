#!/bin/bash
# lab (round 5): row statistics of the LayerNorm fold combined once per producer launch (default from 8,192 rows) against the
# in-tile combine (LTR_STATS_COMB_MIN=1000000000), alternating on one box, both BASELINE workloads; then parity.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-stats_comb}; mkdir -p $O
cd $R
COMMON="--no-cpu-baseline --no-unfused --no-strong --no-scale-points --no-class-head --steady-new 0"
line() { python - "$1" <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = o["kernels"]
print(f"{o['value']:9.0f} req/s  {o['ms_per_step']:8.2f} ms  gemm {k['gemm']['ms_per_step']:8.2f} ms  attn {k['attn']['ms_per_step']:7.2f} ms")
PY
}
{
  for v in 1000000000 8192 1000000000 8192; do
    LTR_STATS_COMB_MIN=$v python bench.py --steps 5 --warmup 2 $COMMON > $O/b125_$v.json 2>/dev/null; echo -n "125m sharegpt  COMB_MIN=$v: "; line $O/b125_$v.json
  done
  for v in 1000000000 8192 1000000000 8192; do
    LTR_STATS_COMB_MIN=$v python bench.py --model 350m --profile lmsys --steps 2 --warmup 1 $COMMON > $O/b350_$v.json 2>/dev/null; echo -n "350m lmsys     COMB_MIN=$v: "; line $O/b350_$v.json
  done
  for v in 1000000000 8192; do
    LTR_STATS_COMB_MIN=$v python bench.py --weight-dtype f16-1pass --steps 5 --warmup 2 $COMMON > $O/b1p_$v.json 2>/dev/null; echo -n "125m one-pass  COMB_MIN=$v: "; line $O/b1p_$v.json
  done
  echo "--- parity"
  python -m pytest tests/test_gpu_scorer.py -x -q -m gpu -k "row_statistics or golden_true or fold" 2>&1 | tail -2
  python -m pytest tests/test_gpu_full_configs.py tests/test_gpu_config1.py tests/test_gpu_outlier.py -x -q -m gpu 2>&1 | tail -2
} > $O/stats_comb.txt 2>&1
cat $O/stats_comb.txt
