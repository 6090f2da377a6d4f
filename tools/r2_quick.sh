#!/bin/bash
# parity suite + kernel durations of the config-3 matvec (channel 1) + HIP-event matvec time
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
WL=${WL:-cfg3_256c3_thick6z} CH=1 bash tools/prof.sh tools/pmc5.py 2>&1 | grep -i "splat2<\|pull_conv2"
WL=${WL:-cfg3_256c3_thick6z} python tools/mv_time.py | tail -3
