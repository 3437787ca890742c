timeout 1500 python -m pytest tests/test_train_step.py -x -q 2>&1 | tail -5
for w in 0 384 512 256; do echo "SPLITK_WGS=$w"; LTR_TRAIN_SPLITK_WGS=$w python bench.py --train --train-slate 32 --train-precision split --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c160-260; done
LTR_TRAIN_SPLITK_WGS=0 python bench.py --train --train-slate 128 --train-precision split --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c160-260
python bench.py --train --train-slate 128 --train-precision split --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c160-260
