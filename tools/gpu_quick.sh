python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_tn" 2>&1 | tail -2
python bench.py --model large --n-passages 40 --batch 8 --breakdown --no-cpu-baseline --all-valid-steps 0 --steps 8 --warmup 3 2>gpurun_out/c4b.txt | cut -c1-220; grep -E "gemm_tn|sum of" gpurun_out/c4b.txt
python bench.py --breakdown --no-cpu-baseline --all-valid-steps 0 --steps 10 --warmup 5 2>gpurun_out/c2b.txt | cut -c1-200; grep -E "gemm_tn|sum of" gpurun_out/c2b.txt
