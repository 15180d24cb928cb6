python -m pytest tests/test_parity_gpu.py tests/test_real_size_gpu.py -m gpu -q -x -k "encoder_space or generate or bf16 or determinism or checkpoint" 2>&1 | tail -2
