set -x
OUT=gpurun_out/r03t
mkdir -p $OUT
for i in 1 2 3 4 5 6; do
  ( time timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "oracle_tokens" ) > $OUT/pytest_$i.log 2>&1
  tail -3 $OUT/pytest_$i.log | head -1
done
( time timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_kernels_gpu.py tests/test_index.py -m gpu -x -q ) > $OUT/pytest_all.log 2>&1
tail -3 $OUT/pytest_all.log
