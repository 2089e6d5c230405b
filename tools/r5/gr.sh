#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3)
T=${GR_TIMEOUT:-1500}
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
