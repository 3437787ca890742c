timeout 1500 python -m pytest tests/test_train_step.py -x -q 2>&1 | tail -5
for w in 0 1; do echo "NARROW=$w"; LTR_TRAIN_SPLITK_NARROW=$w python bench.py --train --train-slate 32 --train-precision split --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c200-245; done
for w in 0 1; do echo "NARROW=$w"; LTR_TRAIN_SPLITK_NARROW=$w python bench.py --train --train-slate 32 --train-precision split --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c200-245; done
