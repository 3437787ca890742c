set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_scorer.py tests/test_gpu_full_configs.py -m gpu -x -q -s 2>&1 | tail -25
for v in 0 1 0 1; do LTR_NO_LASTQ=$v timeout 300 python bench.py --steps 4 --warmup 1 --no-unfused 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('NO_LASTQ=$v', d['value'], d['ms_per_step'], {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('kernels',{}).items()})
"; done
