set -x
OUT=gpurun_out/r03s
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_index.py tests/test_abi.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/pytest.log; tail -2 $OUT/smoke.log; cut -c1-300 $OUT/bench.json
