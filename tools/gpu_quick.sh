python -m pytest tests/test_real_size_gpu.py -q -k "fp8 or batch16 or c4" 2>&1 | grep -E "^E|passed|failed" | head
python bench.py --fp8 --steps 15 --warmup 5 --no-cpu-baseline --all-valid-steps 0 --breakdown 2> gpurun_out/fp8_breakdown.txt | cut -c1-300; head -8 gpurun_out/fp8_breakdown.txt
python bench.py --steps 15 --warmup 5 --no-cpu-baseline --all-valid-steps 0 | cut -c1-200
