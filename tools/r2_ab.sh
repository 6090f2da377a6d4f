#!/bin/bash
# splat2 durations per channel of config 3 (A/B runs)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for c in 0 1 2; do echo "ch $c"; WL=cfg3_256c3_thick6z CH=$c bash tools/prof.sh tools/pmc5.py 2>&1 | grep "splat2<\|pull_conv2"; done
