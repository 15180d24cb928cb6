set -x
OUT=gpurun_out/r03c
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -x -q --durations=8 -k "attention or oracle_tokens" ) > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -25 $OUT/pytest_new.log
for p in 0 1; do
  LAKO_ATTN_PERSIST=$p python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 --breakdown > $OUT/bench_persist$p.json 2> $OUT/bench_persist$p.err
  cut -c1-200 $OUT/bench_persist$p.json; grep -E "attn|sum of" $OUT/bench_persist$p.err
done
