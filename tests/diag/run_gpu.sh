O=gpurun_out/r03_k; mkdir -p $O
python -m pytest tests/test_train_step.py -m gpu -q -s > $O/t_train.log 2>&1; grep -E "^FAILED|fit:|passed|failed|AssertionError|ListMLE" $O/t_train.log | tail -12
python bench.py --train --steps 3 --warmup 1 2>/dev/null | tee $O/train_bench2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1),'ms', round(d['exact_f32_ms_per_step'],1), round(d['value']), 'tok/s', round(d['algorithmic_tflops'],1), 'TFLOP/s')"
python bench.py --train --train-slate 128 --steps 3 --warmup 1 2>/dev/null | tee -a $O/train_bench2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1),'ms', round(d['exact_f32_ms_per_step'],1), round(d['value']), 'tok/s', round(d['algorithmic_tflops'],1), 'TFLOP/s')"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --train --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py $(find $GRAFT_REPO_ROOT/$O/prof -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/train_kernel_stats.csv | head -14
find $GRAFT_REPO_ROOT/$O -name "*.db" -delete
