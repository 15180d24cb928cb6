python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "rmsnorm" 2>&1 | tail -2
