#!/bin/bash
cd $GRAFT_REPO_ROOT
k() { for wl in $WLS; do bash tools/r6_k.sh $wl "${CHS:-1}" | grep "splat2<\|^--"; done; }
echo "== product"; CHS="0 1" WLS="demo_181c3_thick4xyz cfg3_256c3_thick6z cfg2_181c3_1mm" k
echo "== wave-major slots"; UNIRES_LIB=$PWD/build/ab/wm.so CHS="0 1" WLS="demo_181c3_thick4xyz cfg3_256c3_thick6z" k
UNIRES_LIB=$PWD/build/ab/wm.so WLS="cfg4_384c4_iso2" k
UNIRES_LIB=$PWD/build/ab/wm.so python tools/r6_hash.py 2>&1 | tail -3
python tools/r6_hash.py 2>&1 | tail -3
