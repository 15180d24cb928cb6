# round 3, first GPU call: the new parity tests, the GEMM traffic counters of the current kernels, a baseline bench line of this box
set -x
OUT=gpurun_out/r03a
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_xattn_gpu.py tests/test_kernels_gpu.py tests/test_real_size_gpu.py tests/test_parity_gpu.py -m gpu -x -q --durations=15 \
    -k "xattn or scores_and_context or softmax_fwd_bwd or headbatch or reassociated or decode_step or cross_entropy or mx_quantize or fact_scores or c5_ or fp8 or oracle_tokens" ) > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -25 $OUT/pytest_new.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rows in 48000 64000; do
  LAKO_PROBE_TOKENS=$rows bash tools/run_pmc.sh $OUT/pmc_traffic_$rows tools/gemm_probe.py fetch write
  python tools/gemm_traffic.py $OUT/pmc_traffic_$rows $rows > $OUT/gemm_traffic_$rows.json || tail -5 $OUT/pmc_traffic_$rows/fetch.log
  find $OUT/pmc_traffic_$rows -name "*.csv" -size +2M -delete
done
head -c 600 $OUT/gemm_traffic_48000.json
python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 --breakdown > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json; tail -30 $OUT/bench.err
