python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -3
python tools/attn_time.py
export LAKO_LIB=$PWD/lako_amd/liblako_hip_exp.so
LAKO_ATTN_DEBUG=512 TAG=dq-only python tools/attn_time.py
LAKO_ATTN_DEBUG=560 TAG=dq-only-no-staging-no-compute python tools/attn_time.py
LAKO_ATTN_DEBUG=256 TAG=dkv-only python tools/attn_time.py
