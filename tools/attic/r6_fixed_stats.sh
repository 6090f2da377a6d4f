#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2b -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --no-tol-leg --serial-channels --admm-iters 5 > $OUT/bench_prof_serial_fixed.log 2>&1
cp /tmp/kt2b/k_kernel_stats.csv $OUT/r06_bench_serial_fixed_kernel_stats.csv
grep '^{"metric"' $OUT/bench_prof_serial_fixed.log | tail -1 > $OUT/r06_bench_serial_fixed.json
head -8 /tmp/kt2b/k_kernel_stats.csv | cut -c1-140
python -c "
import json; d=json.load(open('$OUT/r06_bench_serial_fixed.json')); print(d['value'], d['roofline']['us_per_launch'], d['subjects_per_sec_tol1e-3'])"
