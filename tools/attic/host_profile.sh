#!/bin/bash
# tools/host_profile.py under a few runtime settings -> gpurun_out/host_profile.txt
mkdir -p gpurun_out
out=gpurun_out/host_profile.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" N=40 timeout 200 python tools/host_profile.py 2> gpurun_out/host_profile.err | python -c "
import json,sys
d=json.load(sys.stdin)
for k in ('tol=0','tol=0.001'):
    for c in ('sleep','spin'):
        r=d[k][c]
        print(k,c,'wall %.2f process %.2f main %.2f others %.2f threads %d'%(r['wall_ms'],r['process_cpu_ms'],r['main_thread_cpu_ms'],r['other_threads_cpu_ms'],r['n_threads']), r['threads_cpu_ms'])
" >> $out
}
if [ -n "$SETS" ]; then
  for e in $SETS; do run $(echo $e | tr "," " "); done
else
  run A=0
  run HSA_ENABLE_INTERRUPT=0
  run ROC_ACTIVE_WAIT_TIMEOUT=0
  run GPU_MAX_HW_QUEUES=2
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
fi
cat $out
