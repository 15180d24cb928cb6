python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "adam or optim or train_steps or overfit" 2>&1 | tail -2
python bench.py --breakdown --no-cpu-baseline --all-valid-steps 0 --steps 10 --warmup 5 2>&1 >/dev/null | grep -E "adamw|sum of"
python bench.py --model large --n-passages 40 --batch 8 --breakdown --no-cpu-baseline --all-valid-steps 0 --steps 5 --warmup 3 2>&1 >/dev/null | grep -E "adamw|sum of"
