#!/bin/bash
cd $GRAFT_REPO_ROOT
k() { for wl in $WLS; do bash tools/r6_k.sh $wl "${CHS:-1}" | grep "splat2<\|^--"; done; }
echo "== product"; WLS="cfg3_256c3_thick6z cfg4_384c4_iso2 demo_181c3_thick4xyz" k
echo "== nontemporal schedule streams (nts)"; UNIRES_LIB=$PWD/build/ab/nts.so WLS="cfg3_256c3_thick6z cfg4_384c4_iso2" k
echo "== UNIRES_S2_YFAST=1"; UNIRES_S2_YFAST=1 WLS="cfg3_256c3_thick6z cfg4_384c4_iso2 demo_181c3_thick4xyz" k
for nb in 928 896 768 744 640 512; do echo "== UNIRES_SPLAT2_BLOCKS=$nb"; UNIRES_SPLAT2_BLOCKS=$nb CHS="0 1" WLS="demo_181c3_thick4xyz" k; UNIRES_SPLAT2_BLOCKS=$nb WLS="cfg3_256c3_thick6z" k; done
