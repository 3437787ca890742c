timeout 1500 python -m pytest tests/test_gpu_attention.py -x -q -s 2>&1 | tail -15
