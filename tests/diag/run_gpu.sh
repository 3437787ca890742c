O=gpurun_out/r03_j; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/full.log 2>&1; tail -5 $O/full.log
python bench.py --model 350m --profile lmsys --no-cpu-baseline --no-strong --no-class-head --steps 3 --warmup 1 > $O/bench_config3.json 2>> $O/bench.err; python -c "
import json
d=json.loads(open('$O/bench_config3.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],1), d['p50_steady_new_latency_ms'], {k:(round(v.get('ms_per_step') or 0,2), v.get('launches_per_step')) for k,v in d['kernels'].items()}, d['roofline']['unfused'])"
