#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of the matvec of workload $1, channels $2 (default "0 1 2")
cd $GRAFT_REPO_ROOT
for ch in ${2:-0 1 2}; do
  echo "-- $1 channel $ch"
  WL=$1 CH=$ch bash tools/prof.sh tools/pmc5.py 2>&1 | grep "unires::k_" | grep -v "build\|plan\|compact" | cut -c1-120
done
