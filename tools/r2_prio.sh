#!/bin/bash
cd $GRAFT_REPO_ROOT
for p in 0 1 2 1 2; do echo "prio_rot $p"; UNIRES_S2_PRIO=$p WL=cfg3_256c3_thick6z CH=1 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "splat2<"; done
