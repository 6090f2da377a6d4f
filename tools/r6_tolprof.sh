#!/bin/bash
# kernel account of y-updates under the reference's stopping rule (tolerance 1e-3, 'max_gain' = the guarded rule),
# channels on streams (default) and one after the other: wall per y-update, kernel time by name, gaps
cd /tmp && export TMPDIR=/tmp
for cs in auto serial; do
  echo "== channel streams: $cs"
  rm -rf /tmp/kt; CS=$cs NUP=8 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/tools/r6_tol.py 2>&1 | grep "ms per"
  python $GRAFT_REPO_ROOT/tools/kstats2b.py /tmp/kt/k_kernel_trace.csv 10 | cut -c1-130
  python $GRAFT_REPO_ROOT/tools/kgaps.py /tmp/kt/k_kernel_trace.csv
done
