#!/bin/bash
# rows per workgroup of the window pull (TI x TJ): kernel durations on several configurations
cd $GRAFT_REPO_ROOT
cp unires_amd/libunires_hip.so /tmp/lib_keep.so
for cfg in ${CFGS:-"8 8" "4 8"}; do set -- $cfg
  UNIRES_HIPCC_EXTRA="-DUNIRES_P2_TI=$1 -DUNIRES_P2_TJ=$2" python __graft_entry__.py --force > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  for w in "cfg2_181c3_1mm 0" "cfg2_181c3_1mm 1" "cfg4_384c4_iso2 0" "cfg4_384c4_iso2 2" "demo_181c3_thick4xyz 2"; do set -- $cfg; set -- $1 $2 $w
    echo "TI=$1 TJ=$2 $3 ch $4"; WL=$3 CH=$4 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "pull_conv2"; done
done
cp /tmp/lib_keep.so unires_amd/libunires_hip.so
