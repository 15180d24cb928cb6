# final confirmation of the tree at the end of round 3: full GPU suite, smoke, the default bench line
set -x
OUT=gpurun_out/r03p
mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/pytest.log; tail -2 $OUT/smoke.log; cut -c1-400 $OUT/bench.json
