#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export WL=cfg3_256c3_thick6z CH=${CH:-1}
for d in 0 1 2 4 3 6 7; do echo "== UNIRES_P2_DBG=$d"; UNIRES_P2_DBG=$d bash tools/prof.sh tools/pmc5.py 2>&1 | grep "pull_conv2"; done > $OUT/abl_pull.log 2>&1
cat $OUT/abl_pull.log
bash tools/pmc2.sh tools/pmc5.py > $OUT/pmc2.log 2>&1; grep -A20 "pull_conv2" $OUT/pmc2.log
