#!/bin/bash
# kernel durations of one matvec per channel for the other configurations
cd $GRAFT_REPO_ROOT
for wl in ${WLS:-demo_181c3_thick4xyz cfg3_256c3_thick6xyz cfg2_181c3_1mm cfg4_384c4_iso2 cfg4_384c4_iso2_gauss}; do
  for ch in ${CHS:-0 1 2}; do
    echo "== $wl ch $ch"; WL=$wl CH=$ch bash tools/prof.sh tools/pmc5.py 2>&1 | grep -v "build\|rigid\|tensor\|\[" | grep "n=  20"
  done
done
