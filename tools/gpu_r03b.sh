set -x
OUT=gpurun_out/r03b
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_real_size_gpu.py tests/test_parity_gpu.py -m gpu -x -q --durations=8 -k "c5_ or fp8 or oracle_tokens" ) > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -25 $OUT/pytest_new.log
python tools/bench_ops.py --only gemm --variants 2,6 --vendor --iters 20 > $OUT/bench_ops_gemm.log 2>&1
cat $OUT/bench_ops_gemm.log
