python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn or enc or dropout" 2>&1 | tail -2
python tools/attn_time.py 2>&1 | tail -1
