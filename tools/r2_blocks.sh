#!/bin/bash
cd $GRAFT_REPO_ROOT
export WL=cfg3_256c3_thick6z CH=${CH:-1}
for b in 1024 768 576 1152 512; do echo "== UNIRES_SPLAT2_BLOCKS=$b"; UNIRES_SPLAT2_BLOCKS=$b bash tools/prof.sh tools/pmc5.py 2>&1 | grep "splat2<"; done
