#!/bin/bash
# Regenerates everything under profiles/r02_* (run on the GPU box; outputs land in gpurun_out/r02)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.log 2>&1; grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/r02_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --admm-iters 5 > $OUT/bench_prof_default.log 2>&1
cp /tmp/kt1/k_kernel_stats.csv $OUT/r02_bench_kernel_stats.csv
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --serial-channels --admm-iters 5 > $OUT/bench_prof_serial.log 2>&1
cp /tmp/kt2/k_kernel_stats.csv $OUT/r02_bench_serial_kernel_stats.csv
grep '^{"metric"' $OUT/bench_prof_serial.log | tail -1 > $OUT/r02_bench_serial.json
cd $GRAFT_REPO_ROOT
# PMC: traffic calibration on known-bytes kernels, then the matvec kernels of every channel
bash tools/traffic2.sh calib -- $GRAFT_REPO_ROOT/tools/mb_traffic > $OUT/r02_traffic_calibration.jsonl 2>$OUT/traffic_calib.err
for c in 0 1 2; do CH=$c WL=cfg3_256c3_thick6z bash tools/traffic2.sh ch$c -- python $GRAFT_REPO_ROOT/tools/pmc5.py; done > $OUT/r02_traffic_pmc.jsonl 2>$OUT/traffic.err
for c in 0 1 2; do echo "== channel $c"; CH=$c WL=cfg3_256c3_thick6z bash tools/pmc2.sh tools/pmc5.py; done > $OUT/r02_sq_counters.txt 2>&1
bash tools/r2_configs.sh > $OUT/configs.txt 2>&1; cp gpurun_out/r2/configs.jsonl $OUT/r02_configs.jsonl
bash tools/r2_timeline.sh > $OUT/r02_splat2_timeline.txt 2>&1
bash tools/r2_timeline_pull.sh > $OUT/r02_pull2_timeline.txt 2>&1
ls -la $OUT
