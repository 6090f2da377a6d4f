#!/bin/bash
# SQ counters (two passes) of the config-3 matvec kernels, channel CH; ablation of splat2 with an -DUNIRES_ABLATE build
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export WL=cfg3_256c3_thick6z CH=${CH:-1}
bash tools/pmc2.sh tools/pmc5.py > $OUT/pmc2.log 2>&1; grep -A20 "k_splat2<\|k_pull_conv2" $OUT/pmc2.log
if [ -n "$ABL" ]; then
  cp unires_amd/libunires_hip.so /tmp/lib_keep.so
  UNIRES_HIPCC_EXTRA=-DUNIRES_ABLATE python __graft_entry__.py --force > /tmp/abl_build.log 2>&1
  for d in 0 1 2 4 8 12; do echo "== UNIRES_S2_DBG=$d"; UNIRES_S2_DBG=$d bash tools/prof.sh tools/pmc5.py 2>&1 | grep "splat2<"; done > $OUT/abl.log 2>&1
  cat $OUT/abl.log
  cp /tmp/lib_keep.so unires_amd/libunires_hip.so
fi
