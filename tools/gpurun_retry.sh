#!/usr/bin/env bash
# Local helper (runs in the build container, not on the GPU box): call gpurun, retrying while no box / slot is free (exit code 3).
#   tools/gpurun_retry.sh <tag> <timeout-seconds> '<command>'     → gpurun_out/<tag>_call.log
tag=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > gpurun_out/${tag}_call.log 2>&1
  rc=$?
  [ $rc -ne 3 ] && break
  sleep 90
done
echo "gpurun rc=$rc" >> gpurun_out/${tag}_call.log
exit $rc
