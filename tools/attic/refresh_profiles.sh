#!/bin/bash
# Regenerates everything under profiles/r01_* (run on the GPU box; outputs land in gpurun_out/r01)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r01; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.log 2>&1; grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/r01_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants > $OUT/bench_prof_default.log 2>&1
cp /tmp/kt1/k_kernel_stats.csv $OUT/r01_bench_kernel_stats.csv
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --serial-channels > $OUT/bench_prof_serial.log 2>&1
cp /tmp/kt2/k_kernel_stats.csv $OUT/r01_bench_serial_kernel_stats.csv
grep '^{"metric"' $OUT/bench_prof_serial.log | tail -1 > $OUT/r01_bench_serial.json
rm -rf /tmp/kt3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --serial-channels --workload cfg3_256c3_thick6z_aligned > $OUT/bench_prof_aligned.log 2>&1
cp /tmp/kt3/k_kernel_stats.csv $OUT/r01_bench_aligned_serial_kernel_stats.csv
grep '^{"metric"' $OUT/bench_prof_aligned.log | tail -1 > $OUT/r01_bench_aligned_serial.json
cd $GRAFT_REPO_ROOT
for c in 0 1 2; do echo channel $c; CH=$c WL=cfg3_256c3_thick6z bash tools/traffic.sh tools/pmc5.py; done > $OUT/r01_traffic_pmc.txt 2>&1
WL=cfg3_256c3_thick6z_aligned bash tools/traffic.sh tools/pmc5.py > $OUT/r01_traffic_aligned_pmc.txt 2>&1
WL=cfg3_256c3_thick6z bash tools/pmc.sh tools/pmc5.py > $OUT/r01_sq_counters.txt 2>&1
WL=cfg3_256c3_thick6z_aligned bash tools/pmc.sh tools/pmc5.py >> $OUT/r01_sq_counters.txt 2>&1
ls -la $OUT
