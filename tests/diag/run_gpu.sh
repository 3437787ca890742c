timeout 1500 python -m pytest tests/test_train_step.py -x -q 2>&1 | tail -15
python bench.py --train --train-slate 32 --train-precision split --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-300
