python -m pytest tests/test_parity_gpu.py tests/test_drivers_gpu.py -m gpu -q -x -k "generate or decode or collated or greedy or driver" 2>&1 | tail -2
