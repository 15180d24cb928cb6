set -x
OUT=gpurun_out/r03h
mkdir -p $OUT
( time timeout 2400 python -m pytest tests/test_retriever.py tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_drivers_gpu.py tests/test_checkpoint_eval.py -m gpu -q --maxfail=8 --durations=8 -k "retriever or tile_queue or oracle_tokens or train_retriever or tile_variants" ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
for q in 0 1; do
  LAKO_TUNING=gemm_nt_queue=$q python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 --breakdown > $OUT/bench_queue$q.json 2> $OUT/bench_queue$q.err
  cut -c1-200 $OUT/bench_queue$q.json; head -3 $OUT/bench_queue$q.err
done
