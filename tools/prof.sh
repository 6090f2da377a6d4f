#!/bin/bash
# usage: tools/prof.sh <script.py>  -> true kernel durations via rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/$1 > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/kt/k_kernel_trace.csv || tail -5 /tmp/kt.log
