#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
cp unires_amd/libunires_hip.so /tmp/lib_keep.so
UNIRES_HIPCC_EXTRA=-DUNIRES_S2_PROF python __graft_entry__.py --force > /tmp/prof_build.log 2>&1 || tail /tmp/prof_build.log
UNIRES_S2_PROF_OUT=$OUT/s2_timeline.txt WL=${WL:-cfg3_256c3_thick6z} CH=${CH:-1} python tools/pmc5.py > /dev/null 2>&1
python tools/s2_timeline.py $OUT/s2_timeline.txt
cp /tmp/lib_keep.so unires_amd/libunires_hip.so
