#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/f1; mkdir -p $OUT
{
WL=cfg2_181c3_1mm SAVE=/tmp/ref.pt UNIRES_NO_ATA1=1 NOTIME=1 timeout 600 python tools/f1_check.py 2>&1 | grep -v "info\|amdgpu"
for q in 0 1; do
  for wl in cfg2_181c3_1mm dn_256c3_1mm; do
  echo "== UNIRES_F1_QUEUE=$q $wl"
  if [ $wl = cfg2_181c3_1mm ]; then C=/tmp/ref.pt; else C=; fi
  UNIRES_F1_QUEUE=$q WL=$wl CMP=$C timeout 600 python tools/f1_check.py 2>&1 | grep -v "info\|amdgpu"
  done
done
} > $OUT/queue.txt 2>&1
cat $OUT/queue.txt
timeout 900 python -m pytest tests/test_gpu_ata1.py tests/test_gpu_path.py tests/test_gpu_cg.py -x -q -m gpu -k "dn or ata1 or single or cg" 2>&1 | tail -3
