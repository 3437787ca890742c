#!/bin/bash
# Queue-size sweep of the cold ranker call on one GPU; writes JSON lines.
out=${1:-gpurun_out/queue_sweep.jsonl}
: > $out
for q in 256 1024 2048 4096 8192 16384 32768 65536; do
  steps=3; [ $q -ge 32768 ] && steps=2
  python bench.py --steps $steps --warmup 1 --queue $q --no-cpu-baseline --steady-new 256 >> $out
done
python - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print(f"N={d['config']['queue_per_gpu']:6d} T={d['config']['tokens_total']:8d}  {d['value']:9.0f} req/s  cold p50 {d['p50_rank_latency_ms']:8.2f} ms  steady {d['p50_steady_rank_latency_ms']*1e3:7.1f} us  gemm {d['roofline']['achieved']:.0f} TF")
PY
