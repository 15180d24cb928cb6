set -x
OUT=gpurun_out/r03j
mkdir -p $OUT
export LAKO_LIB=$PWD/lako_amd/liblako_hip_exp.so
for cfg in "old 1024" "new 512" "new 768" "new 1024" "new 1536" "new 2048" "old 1024" "new 768"; do
  set -- $cfg
  o=0; [ $1 = old ] && o=1
  echo "=== $1 grid $2" >> $OUT/norm.txt
  LAKO_RMS_OLD=$o LAKO_RMS_GRID=$2 python tools/bench_ops.py --only norm 2>&1 | grep -v amdgpu >> $OUT/norm.txt
done
cat $OUT/norm.txt
unset LAKO_LIB
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=8 -k "rmsnorm or norm" ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 --breakdown > $OUT/bench.json 2> $OUT/bench.err
cut -c1-260 $OUT/bench.json; head -12 $OUT/bench.err
