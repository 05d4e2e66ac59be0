#!/bin/bash
# gpurun with retries while the pool is busy (exit code 3 = no box / slot free, nothing charged):
#     scripts/gpurun_retry.sh <timeout-seconds> '<command>' > gpurun_out/<log> 2>&1
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
