python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "long_answers" 2>&1 | tail -5
