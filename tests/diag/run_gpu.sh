O=gpurun_out/r03_f; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/full.log 2>&1; tail -8 $O/full.log
python bench.py --no-cpu-baseline --no-strong > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['p50_steady_new_latency_ms'], d['roofline']['achieved'], {k:v.get('ms_per_step') for k,v in d['kernels'].items()})
PY
