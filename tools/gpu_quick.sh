python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -2
python tools/attn_time.py
NODROP=1 TAG=nodrop python tools/attn_time.py
