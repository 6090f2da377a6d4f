#!/bin/bash
# channels on streams of their own with the matvec kernels' occupancy reduced (room for the other channels' vector
# kernels): bench value per setting -> gpurun_out/overlap_scan.txt
mkdir -p gpurun_out; out=gpurun_out/overlap_scan.txt; : > $out
run() {
  env "$@" python bench.py --workload ${WL:-cfg3_256c3_thick6z} --no-cpu-baseline --no-variants $MODE --admm-iters 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.load(sys.stdin); print('%-70s it/s %6.0f  ms/step %7.3f  matvec %6.1f us  subj/s %.3f' % ('$*', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['subjects_per_sec']))" >> $out
}
MODE=""; run A=serial
MODE="--channel-streams"
for e in $SETS; do run $(echo $e | tr ',' ' '); done
cat $out
