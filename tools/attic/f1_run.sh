#!/bin/bash
# first contact of the single-pass AtA kernel with the hardware: parity cases, bit-compare with the pair, timings per tile shape
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/f1; mkdir -p $OUT
export UNIRES_ATA1_VERBOSE=1
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_orient.py tests/test_golden.py -x -q -m gpu -k "dn or denois or R1 or pull" 2>&1 | tail -15 > $OUT/pytest_dn.txt
WL=cfg2_181c3_1mm SAVE=/tmp/ref.pt UNIRES_NO_ATA1=1 timeout 600 python tools/f1_check.py > $OUT/pair.txt 2>&1
for tile in ${TILES:-44}; do
  UNIRES_F1_TILE=$tile WL=cfg2_181c3_1mm CMP=/tmp/ref.pt timeout 600 python tools/f1_check.py > $OUT/tile_$tile.txt 2>&1
done
cat $OUT/pytest_dn.txt $OUT/pair.txt $OUT/tile_*.txt
