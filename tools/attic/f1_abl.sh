#!/bin/bash
# phase ablation of k_ata1 (needs a -DUNIRES_ABLATE build): per-channel matvec times with parts switched off
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/f1; mkdir -p $OUT
for dbg in 0 1 2 3 4 8 16 24 7; do
  echo "== UNIRES_F1_DBG=$dbg"
  UNIRES_F1_DBG=$dbg WL=${WL:-cfg2_181c3_1mm} timeout 600 python tools/f1_check.py 2>&1 | grep "matvec"
done > $OUT/ablate.txt 2>&1
cat $OUT/ablate.txt
