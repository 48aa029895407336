#!/bin/bash
# bash tools/gpu_retry.sh <timeout-seconds> '<command>': gpurun, retried while the pod's GPU slots are busy (exit code 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpurun_last.log 2>&1
  rc=$?
  if ! grep -q "status=transient" /tmp/gpurun_last.log; then break; fi
  sleep 45
done
grep -v "^\[gpurun\] sending\|merged" /tmp/gpurun_last.log | tail -${TAILN:-14}
exit $rc
