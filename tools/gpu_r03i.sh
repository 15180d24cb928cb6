set -x
OUT=gpurun_out/r03i
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_real_size_gpu.py -m gpu -q --maxfail=8 --durations=8 -k "weight_gradient_stream or c2_batch16" ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
for cfg in "0 0" "1 0" "1 1" "0 0" "1 0"; do
  set -- $cfg
  LAKO_DW_STREAM=$1 LAKO_TUNING=gemm_nt_queue=$2 python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 --breakdown > $OUT/bench_dw$1_q$2.json 2> $OUT/bench_dw$1_q$2.err
  cut -c1-260 $OUT/bench_dw$1_q$2.json; head -4 $OUT/bench_dw$1_q$2.err
done
