O=gpurun_out/r03_g; mkdir -p $O
python -m pytest tests/test_gpu_scorer.py -m gpu -x -q -s -k "golden or class or pool_head or hf_written or edge" > $O/t1.log 2>&1; tail -12 $O/t1.log
python -m pytest tests/test_gpu_full_configs.py tests/test_ltr_head.py tests/test_train_step.py -m gpu -x -q -k "config3 or head or round_trips" > $O/t2.log 2>&1; tail -3 $O/t2.log
python bench.py --no-cpu-baseline --no-strong --no-unfused --steady-new 0 --steps 2 --warmup 1 2> $O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernels'].get('class_head'), d['kernels']['pool'])"
python bench.py --model 350m --profile lmsys --no-cpu-baseline --no-strong --no-unfused --steady-new 0 --steps 2 --warmup 1 2>> $O/bench.err | tee $O/bench350.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:(v.get('ms_per_step'), v.get('launches_per_step')) for k,v in d['kernels'].items()})"
tail -3 $O/bench.err
