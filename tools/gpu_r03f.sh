set -x
OUT=gpurun_out/r03f
mkdir -p $OUT
python tools/fp8_bound.py > $OUT/fp8_bound.txt 2>&1; cat $OUT/fp8_bound.txt
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
