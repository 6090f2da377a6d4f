#!/bin/bash
# same-box A/B of two builds of the library: per-kernel durations of workload $1 (channels $2) and the bench line,
# first with build/ab/${BASE:-base_r5}.so (UNIRES_LIB), then with the in-tree library
cd $GRAFT_REPO_ROOT
B="python bench.py --workload $1 --no-cpu-baseline --no-variants --admm-iters 2"
pr() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); r=d['roofline']; print('$1', 'it/s %.0f ms/step %.3f mv %.1f cold %.1f bych %s' % (d['value'], d['ms_per_step'], r['us_per_launch'], r['us_per_launch_cold'], ['%.1f'%v for v in r['us_per_launch_by_channel']]))"; }
for side in base new; do
  if [ $side = base ]; then export UNIRES_LIB=$GRAFT_REPO_ROOT/build/ab/${BASE:-base_r5}.so; else unset UNIRES_LIB; fi
  echo "== $side"
  bash tools/r6_k.sh $1 "${2:-1}"
  [ -z "$NOBENCH" ] && $B 2>/dev/null | pr $side
done
