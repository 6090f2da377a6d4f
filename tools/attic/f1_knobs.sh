#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/f1; mkdir -p $OUT
run() { echo "== $*"; env "$@" WL=cfg2_181c3_1mm timeout 600 python tools/f1_check.py 2>&1 | grep "matvec"; }
{
run A=0
run UNIRES_F1_PRIO=0
run UNIRES_F1_TILE_COST=9
run UNIRES_F1_TILE_COST=14
run UNIRES_F1_BLOCKS=2048
run UNIRES_F1_BLOCKS=512
} > $OUT/knobs.txt 2>&1
cat $OUT/knobs.txt
