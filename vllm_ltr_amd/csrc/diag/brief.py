"""stdin: bench.py JSON line -> one short line (diagnostic helper)."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d["kernels"]
parts = [tag, f"{d['value']:.0f} req/s", f"{d['ms_per_step']:.1f} ms/step", f"steady {d['p50_steady_rank_latency_ms']*1e3:.0f}us"]
for n, v in k.items():
    parts.append(f"{n} {v['ms_per_step']:.1f}ms" + (f" {v['tflops']:.0f}TF" if "tflops" in v else f" {v['gbs']:.0f}GB/s"))
print(" | ".join(parts))
