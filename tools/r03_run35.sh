timeout -k 10 300 python -m pytest tests/test_gpu_tuning.py -m gpu -x -q -k "deferred" 2>&1 | tail -n 15
