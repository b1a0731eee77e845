#!/bin/bash
# tools/gpurun_retry.sh <timeout-seconds> '<command>': retry while the pool says "no slot free" (exit 3: nothing charged)
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
