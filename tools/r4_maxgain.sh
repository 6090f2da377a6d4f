#!/bin/bash
# Round 4, item 2: the reference-default CG mode (tol 1e-3, 'max_gain').  Kernel statistics of 1 + 15 y-updates
# (tools/tol_time.py) under the chunked enqueue and under the full enqueue of rounds 1-3.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 2 0; do
  rm -rf /tmp/mg$k && UNIRES_CG_CHUNK=$k rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mg$k -o k -- python $GRAFT_REPO_ROOT/tools/tol_time.py > $OUT/maxgain_chunk$k.log 2>&1
  cp /tmp/mg$k/k_kernel_stats.csv $OUT/r04_maxgain_chunk${k}_kernel_stats.csv
  tail -1 $OUT/maxgain_chunk$k.log
done
