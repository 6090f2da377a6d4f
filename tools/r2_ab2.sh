#!/bin/bash
# A/B of a build-flag variant: kernel durations per channel of config 3 with the committed build, then with $1
cd $GRAFT_REPO_ROOT
cp unires_amd/libunires_hip.so /tmp/lib_keep.so
for c in 0 1 2; do echo "base ch $c"; WL=cfg3_256c3_thick6z CH=$c bash tools/prof.sh tools/pmc5.py 2>&1 | grep "splat2<\|pull_conv2"; done
UNIRES_HIPCC_EXTRA=$1 python __graft_entry__.py --force > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
for c in 0 1 2; do echo "$1 ch $c"; WL=cfg3_256c3_thick6z CH=$c bash tools/prof.sh tools/pmc5.py 2>&1 | grep "splat2<\|pull_conv2"; done
python -m pytest tests/test_golden.py tests/test_gpu_sizes.py -m gpu -x -q 2>&1 | tail -2
cp /tmp/lib_keep.so unires_amd/libunires_hip.so
