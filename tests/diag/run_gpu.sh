O=gpurun_out/r03_l; mkdir -p $O
for cfg in "1024 3072" "0 100000000"; do set -- $cfg; echo "SMALL_M=$1 MID_M=$2"; LTR_GEMM_SMALL_M=$1 LTR_GEMM_MID_M=$2 python bench.py --no-cpu-baseline --no-strong --no-unfused --no-class-head --steady-new 0 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), {k:round(v.get('ms_per_step') or 0,2) for k,v in d['kernels'].items()}, round(d['roofline']['achieved'],1))"; done | tee $O/mid_tiles_full_size.txt
