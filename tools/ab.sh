#!/bin/bash
# A/B loop on the GPU box: every line of the variant file is  name | hipcc flags | env assignments | command
# The library is rebuilt (forced) whenever the flags differ from the previous line's; the command runs
# with the env assignments; its last lines are printed under the variant's name.
#   usage: tools/ab.sh <variant file>
cd $GRAFT_REPO_ROOT
last="__none__"
b() { python bench.py --no-cpu-baseline --no-variants --admm-iters 2 "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s it/s %8.0f  ms/step %7.3f  matvec %7.1f us (cold: graph %.1f eager %.1f)  frac %.3f' % (d['config']['workload'], d['value'], d['ms_per_step'], r['us_per_launch'], r['us_per_launch_graph'], r['us_per_launch_eager'], r['frac']))"; }
k() { WL=$1 CH=${2:-1} bash tools/prof.sh tools/pmc5.py 2>&1 | grep "unires::k_" | grep -v "build\|plan" | cut -c1-110; }
export -f b k
while IFS='|' read -r name flags envs cmd; do
  [ -z "$name" ] && continue
  case "$name" in \#*) continue;; esac
  if [ "$flags" != "$last" ]; then
    UNIRES_HIPCC_EXTRA="$flags" python __graft_entry__.py --force > /tmp/build.log 2>&1 || { echo "== $name: BUILD FAILED"; tail -5 /tmp/build.log; last="__none__"; continue; }
    last="$flags"
  fi
  echo "== $name   [$flags] [$envs]"
  env $envs bash -c "$cmd" 2>&1 | tail -12
done < "$1"
# leave the default build behind
python __graft_entry__.py --force > /dev/null 2>&1
