timeout 1500 python -m pytest tests/test_train_step.py -x -q 2>&1 | tail -5
python bench.py --train --train-slate 32 --train-precision split --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_gemm_epilogue.py tests/test_gpu_small_batches.py -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-unfused --no-strong --no-class-head --steady-new 0 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['kernels']['gemm'])"
