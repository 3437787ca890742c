O=gpurun_out/r03_k; mkdir -p $O
python -m pytest tests/test_train_step.py -m gpu -q -s > $O/t_train.log 2>&1; grep -E "^FAILED|fit:|passed|failed|AssertionError|ListMLE" $O/t_train.log | tail -12
for f in 1 0; do LTR_TRAIN_F32=$f python bench.py --train --steps 3 --warmup 1 2>/dev/null | tee -a $O/train_bench.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('F32=$f', round(d['ms_per_step'],1),'ms', round(d['value']), 'tok/s', round(d['algorithmic_tflops'],1), 'TFLOP/s')"; done
for n in 64 128; do LTR_TRAIN_F32=0 python bench.py --train --train-slate $n --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slate $n', round(d['ms_per_step'],1),'ms', round(d['value']), 'tok/s', round(d['algorithmic_tflops'],1), 'TFLOP/s')"; done
