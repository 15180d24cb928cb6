# scratch: the command list of the current gpurun call (edited per call; see tools/gpu_round.sh for the round-end sequence)
mkdir -p gpurun_out/r02k
( time python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r02k/pytest.log 2>&1; tail -9 gpurun_out/r02k/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py ) > gpurun_out/r02k/bench.json 2> gpurun_out/r02k/bench.err; cut -c1-230 gpurun_out/r02k/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pk -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --all-valid-steps 0 > gpurun_out/r02k/bench_under_rocprof.json 2> gpurun_out/r02k/prof.log
python tools/rocpd_stats.py /tmp/pk/b_results.db > gpurun_out/r02k/bench_kernel_stats.csv
python tools/rocpd_timeline.py /tmp/pk/b_results.db 6 > gpurun_out/r02k/step_timeline.txt; head -30 gpurun_out/r02k/step_timeline.txt
