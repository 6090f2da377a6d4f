#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/f1; mkdir -p $OUT
{
for ex in 0 1; do
  echo "== UNIRES_S2_EXACT=$ex"
  UNIRES_S2_EXACT=$ex UNIRES_SPLAT2_VERBOSE=1 WL=cfg3_256c3_thick6z timeout 600 python tools/mv_time.py 2>&1 | grep -v amdgpu
done
} > $OUT/s2_exact.txt 2>&1
cat $OUT/s2_exact.txt
timeout 1200 python -m pytest tests/test_gpu_path.py tests/test_gpu_ops.py tests/test_gpu_orient.py -x -q -m gpu 2>&1 | tail -3
