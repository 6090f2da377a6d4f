#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/f1; mkdir -p $OUT
export UNIRES_ATA1_VERBOSE=1
{
WL=cfg2_181c3_1mm SAVE=/tmp/ref.pt UNIRES_NO_ATA1=1 NOTIME=1 timeout 600 python tools/f1_check.py 2>&1 | grep -v "info\|amdgpu"
for pack in 1; do
  echo "== UNIRES_F1_PACK=$pack"
  UNIRES_F1_EXACT=${EXACT:-1} UNIRES_F1_PACK=$pack WL=cfg2_181c3_1mm CMP=/tmp/ref.pt timeout 600 python tools/f1_check.py 2>&1 | grep -v "info\|amdgpu"
done
} > $OUT/pack.txt 2>&1
cat $OUT/pack.txt
timeout 900 python -m pytest tests/test_gpu_ata1.py tests/test_gpu_path.py -x -q -m gpu -k "dn or ata1 or single" 2>&1 | tail -3
