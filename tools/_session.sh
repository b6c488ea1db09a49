timeout 900 python -m pytest tests/test_gpu_soak.py tests/test_gpu_wave.py -m gpu -q -x 2>&1 | tail -4
