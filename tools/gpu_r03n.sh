set -x
OUT=gpurun_out/r03n
mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_index.py -m gpu -q --maxfail=8 --durations=5 ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
python tools/bench_ops.py --only index 2>&1 | grep -v amdgpu > $OUT/bench_ops_index.txt; cat $OUT/bench_ops_index.txt
