# scratch: the command list of the current gpurun call (edited per call; see tools/gpu_round.sh for the round-end sequence)
mkdir -p gpurun_out/r02i
( time python -m pytest tests -m gpu -q --durations=6 ) > gpurun_out/r02i/pytest.log 2>&1; tail -12 gpurun_out/r02i/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --fp8 --no-cpu-baseline --all-valid-steps 0 --steps 10 --warmup 4 2>/dev/null | cut -c1-200
( time python bench.py ) > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err; cut -c1-260 gpurun_out/r02i/bench.json
python tools/generate_probe.py 2>&1 | tail -2
