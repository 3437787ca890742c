O=gpurun_out/r03_m; mkdir -p $O
python -m pytest tests/test_gpu_scorer.py tests/test_gpu_small_batches.py tests/test_gpu_gemm_epilogue.py tests/test_gpu_full_configs.py -m gpu -q > $O/t1.log 2>&1; tail -4 $O/t1.log
for i in 1 2; do python bench.py --model 350m --profile lmsys --no-cpu-baseline --no-strong --no-unfused --no-class-head --steady-new 0 --steps 3 --warmup 1 2>> $O/bench.err | tee $O/bench350.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],1), {k:(round(v.get('ms_per_step') or 0,2), v.get('launches_per_step')) for k,v in d['kernels'].items()})"; done
