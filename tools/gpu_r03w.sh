set -x
OUT=gpurun_out/r03w
mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_index.py tests/test_checkpoint_eval.py tests/test_drivers_gpu.py -m gpu -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
