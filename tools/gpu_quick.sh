mkdir -p gpurun_out/r02j
python bench.py --model large --n-passages 40 --batch 8 --no-cpu-baseline --all-valid-steps 0 --steps 15 --warmup 4 > gpurun_out/r02j/bench_c4.json 2>/dev/null; cut -c1-200 gpurun_out/r02j/bench_c4.json
python bench.py --model large --n-passages 100 --batch 8 --no-cpu-baseline --all-valid-steps 0 --steps 8 --warmup 3 > gpurun_out/r02j/bench_c5.json 2>/dev/null; cut -c1-200 gpurun_out/r02j/bench_c5.json
python bench.py --model large --n-passages 100 --batch 8 --fp8 --no-cpu-baseline --all-valid-steps 0 --steps 8 --warmup 3 > gpurun_out/r02j/bench_c5_fp8.json 2>/dev/null; cut -c1-200 gpurun_out/r02j/bench_c5_fp8.json
