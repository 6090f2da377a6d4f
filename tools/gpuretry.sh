#!/bin/bash
# usage: tools/gpuretry.sh <timeout> '<command>'   - retries gpurun while every GPU slot is busy (exit 3)
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $1 -- "$2"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
