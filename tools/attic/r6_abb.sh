#!/bin/bash
# bench-line A/B (base library via UNIRES_LIB, then the in-tree one) over the workloads given as arguments
cd $GRAFT_REPO_ROOT
pr() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); r=d['roofline']; print('%-28s %-5s' % (d['config']['workload'], '$1'), 'it/s %.0f ms/step %.3f mv %.1f cold %.1f frac %.3f bych %s' % (d['value'], d['ms_per_step'], r['us_per_launch'], r['us_per_launch_cold'], r['frac'], ['%.1f'%v for v in r['us_per_launch_by_channel']]))"; }
for wl in "$@"; do
  for side in base new; do
    if [ $side = base ]; then export UNIRES_LIB=$GRAFT_REPO_ROOT/build/ab/${BASE:-base_r5}.so; else unset UNIRES_LIB; fi
    python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 2 2>/dev/null | pr $side
  done
done
