#!/bin/bash
# round-2 iteration loop on the GPU box: parity suite, then matvec timing + kernel trace of config 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
UNIRES_SPLAT2_VERBOSE=1 WL=cfg3_256c3_thick6z python tools/mv_time.py > $OUT/mv_time.log 2>&1; tail -12 $OUT/mv_time.log
WL=cfg3_256c3_thick6z CH=1 bash tools/prof.sh tools/pmc5.py > $OUT/kt.log 2>&1; cat $OUT/kt.log
