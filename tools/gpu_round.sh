# The round-end sequence on an MI355X box (one `gpurun` call): full GPU tests, smoke, the default bench line, a kernel trace of a short
# bench run with the per-kernel table and one step's timeline, and the benches of configs 4 / 5.  Outputs under gpurun_out/$TAG;
# copy what is to be judged into profiles/ (see profiles/README.md).
#   gpurun --timeout 3600 -- 'TAG=r03a bash tools/gpu_round.sh'
set -x
TAG=${TAG:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --all-valid-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
python tools/rocpd_stats.py /tmp/prof_$TAG/b_results.db > $OUT/bench_kernel_stats.csv
python tools/rocpd_timeline.py /tmp/prof_$TAG/b_results.db 6 > $OUT/step_timeline.txt
python bench.py --model large --n-passages 40 --batch 8 --no-cpu-baseline --all-valid-steps 0 --steps 15 --warmup 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --model large --n-passages 100 --batch 8 --no-cpu-baseline --all-valid-steps 0 --steps 8 --warmup 3 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python tools/generate_probe.py > $OUT/generate_probe.txt 2>&1
tail -5 $OUT/pytest.log; cat $OUT/bench.json | cut -c1-300; tail -2 $OUT/generate_probe.txt
