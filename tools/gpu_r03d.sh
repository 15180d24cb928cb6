set -x
OUT=gpurun_out/r03d
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -x -q --durations=8 -k "attention or oracle_tokens" ) > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -12 $OUT/pytest_new.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in 0 1; do
  LAKO_ATTN_PERSIST=$p rocprofv3 --kernel-trace --stats -d /tmp/prof_attn$p -o a -- python3 tools/attn_time.py > $OUT/attn_time_persist$p.txt 2>&1
  python tools/rocpd_stats.py /tmp/prof_attn$p/a_results.db | grep -E "enc_|Name" | cut -c1-160 > $OUT/attn_kernels_persist$p.csv
  tail -1 $OUT/attn_time_persist$p.txt; cat $OUT/attn_kernels_persist$p.csv
done
export LAKO_LIB=$GRAFT_REPO_ROOT/lako_amd/liblako_hip_exp.so
for aux in 0 16 2 18 17; do
  LAKO_TUNING=gemm_nt_store_aux=$aux python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 --breakdown > $OUT/bench_aux$aux.json 2> $OUT/bench_aux$aux.err
  echo "aux $aux: $(cut -c90-200 $OUT/bench_aux$aux.json)"; grep -E "gemm_nt.11|sum of" $OUT/bench_aux$aux.err
done
