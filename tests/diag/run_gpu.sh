B="python bench.py --no-cpu-baseline --no-unfused --no-strong --no-class-head --steady-new 0 --steps 5 --warmup 2"
for i in 1 2; do
for o in 1 2 4 8; do
echo "CHUNK=$o"; LTR_ATTN_XCD_CHUNK=$o $B 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['kernels']['attn']['ms_per_step'], b['kernels']['gemm']['ms_per_step'])"
done; done
LTR_ATTN_XCD_CHUNK=4 timeout 900 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -2
