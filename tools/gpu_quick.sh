mkdir -p gpurun_out/r02d
python bench.py --model large --n-passages 40 --batch 8 --steps 12 --warmup 4 --no-cpu-baseline --all-valid-steps 3 > gpurun_out/r02d/bench_c4.json 2> gpurun_out/r02d/bench_c4.err; cut -c1-250 gpurun_out/r02d/bench_c4.json
python bench.py --model large --n-passages 40 --batch 8 --steps 12 --warmup 4 --no-cpu-baseline --all-valid-steps 0 --fp8 > gpurun_out/r02d/bench_c4_fp8.json 2> gpurun_out/r02d/bench_c4_fp8.err; cut -c1-250 gpurun_out/r02d/bench_c4_fp8.json
python bench.py --model large --n-passages 100 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --all-valid-steps 0 --fp8 > gpurun_out/r02d/bench_c5_fp8.json 2> gpurun_out/r02d/bench_c5_fp8.err; cut -c1-250 gpurun_out/r02d/bench_c5_fp8.json; tail -3 gpurun_out/r02d/bench_c5_fp8.err
python bench.py --model large --n-passages 100 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --all-valid-steps 0 > gpurun_out/r02d/bench_c5_bf16.json 2> gpurun_out/r02d/bench_c5_bf16.err; cut -c1-250 gpurun_out/r02d/bench_c5_bf16.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r02d/prof_c4 -o b -- python3 bench.py --model large --n-passages 40 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --all-valid-steps 0 > gpurun_out/r02d/bench_c4_under_rocprof.json 2> gpurun_out/r02d/prof_c4.log
python tools/rocpd_stats.py $(ls gpurun_out/r02d/prof_c4/*/*.db | head -1) > gpurun_out/r02d/c4_kernel_stats.csv; head -8 gpurun_out/r02d/c4_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/r02d/prof_c4
