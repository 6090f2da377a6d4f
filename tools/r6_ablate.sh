#!/bin/bash
cd $GRAFT_REPO_ROOT
export UNIRES_LIB=$PWD/build/ab/abl.so
for wl in cfg4_384c4_iso2 cfg3_256c3_thick6z; do
echo "== $wl pull ablation (abl build): UNIRES_P2_DBG bits 1 no staging, 2 no sampling, 4 no conv/store"
for dbg in 0 1 2 4 7; do
  echo "-- UNIRES_P2_DBG=$dbg"
  UNIRES_P2_DBG=$dbg WL=$wl CH=1 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "k_pull_conv2"
done
echo "== $wl splat ablation: UNIRES_S2_DBG bits 1 no stream, 2 no epilogue, 4 no LDS updates, 8 no source loads"
for dbg in 0 1 2 3; do
  echo "-- UNIRES_S2_DBG=$dbg"
  UNIRES_S2_DBG=$dbg WL=$wl CH=1 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "k_splat2<"
done
done
