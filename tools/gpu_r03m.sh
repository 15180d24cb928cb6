set -x
OUT=gpurun_out/r03m
mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_index.py tests/test_kernels_gpu.py -m gpu -q --maxfail=8 --durations=8 -k "pq or topk or indexer or rmsnorm" ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
python tools/bench_ops.py --only index 2>&1 | grep -v amdgpu > $OUT/bench_ops_index.txt; cat $OUT/bench_ops_index.txt
bash tools/gpu_r03l.sh
