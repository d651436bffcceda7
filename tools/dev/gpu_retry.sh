#!/bin/bash
# usage: gpu_retry.sh <timeout-seconds> <logfile> <command...>: retry gpurun while all slots are busy (exit code 3)
T=$1; LOG=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if ! grep -q "status=transient" $LOG; then exit $rc; fi
  sleep 90
done
exit 3
