# flakiness insurance: the full GPU suite three times in a row on one box (no -x: every failure is listed)
set -x
OUT=gpurun_out/r03v
mkdir -p $OUT
for i in 1 2 3; do
  ( time timeout 1800 python -m pytest tests -m gpu -q ) > $OUT/pytest_$i.log 2>&1
  grep -h "passed\|failed" $OUT/pytest_$i.log | tail -1
  grep -h "^FAILED" $OUT/pytest_$i.log | head
done
