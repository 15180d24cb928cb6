# scratch: the command list of the current gpurun call (edited per call; see tools/gpu_round.sh for the round-end sequence)
mkdir -p gpurun_out/r02h
( time python -m pytest tests -m gpu -q -x --durations=8 ) > gpurun_out/r02h/pytest.log 2>&1; tail -14 gpurun_out/r02h/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py ) > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err; cut -c1-300 gpurun_out/r02h/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/ph -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --all-valid-steps 0 > gpurun_out/r02h/bench_under_rocprof.json 2> gpurun_out/r02h/prof.log
python tools/rocpd_stats.py /tmp/ph/b_results.db > gpurun_out/r02h/bench_kernel_stats.csv; head -12 gpurun_out/r02h/bench_kernel_stats.csv | cut -c1-150
