#!/bin/bash
# phase ablation of k_splat2 on config 3 (needs a -DUNIRES_ABLATE build): rocprofv3 kernel durations per UNIRES_S2_DBG
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/f1; mkdir -p $OUT
for dbg in 0 1 2 4 8 12; do
  echo "== UNIRES_S2_DBG=$dbg"
  UNIRES_S2_DBG=$dbg WL=cfg3_256c3_thick6z CH=${CH:-1} bash tools/prof.sh tools/pmc5.py 2>&1 | grep "k_splat2<\|k_pull_conv2"
done > $OUT/s2_ablate.txt 2>&1
cat $OUT/s2_ablate.txt
