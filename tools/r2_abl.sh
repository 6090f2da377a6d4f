#!/bin/bash
# splat2 ablation (UNIRES_S2_DBG bits) + SQ counters on config 3, channel CH
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export WL=cfg3_256c3_thick6z CH=${CH:-1}
for d in 0 1 2 4 8 12; do echo "== UNIRES_S2_DBG=$d"; UNIRES_S2_DBG=$d bash tools/prof.sh tools/pmc5.py 2>&1 | grep "splat2<\|pull_conv"; done > $OUT/abl.log 2>&1
cat $OUT/abl.log
bash tools/pmc2.sh tools/pmc5.py > $OUT/pmc2.log 2>&1; cat $OUT/pmc2.log
