#!/bin/bash
# r3 run 1: parity suite, then A/B of the round's first changes (pull staging table, flat A = I
# stencil, CG scalar fold) on configs 3 and 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
b() { python bench.py --no-cpu-baseline --no-variants --admm-iters 5 "$@" 2>>$OUT/bench.err | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s it/s %8.0f  ms/step %7.3f  matvec %7.1f us (graph %.1f eager %.1f)  frac %.3f' % (d['config']['workload'], d['value'], d['ms_per_step'], r['us_per_launch'], r['us_per_launch_graph'], r['us_per_launch_eager'], r['frac']))"; }
echo "== cfg3 default"; b
echo "== cfg3 serial";  b --serial-channels
echo "== cfg3 serial, CG_FOLD=0"; UNIRES_CG_FOLD=0 b --serial-channels
echo "== cfg3 CG_FOLD=0"; UNIRES_CG_FOLD=0 b
echo "== cfg1 flat"; b --workload cfg1_181c1_denoise
echo "== cfg1 lines (NO_FLAT)"; UNIRES_NO_FLAT=1 b --workload cfg1_181c1_denoise
echo "== cfg1 lines, CG_FOLD=0"; UNIRES_NO_FLAT=1 UNIRES_CG_FOLD=0 b --workload cfg1_181c1_denoise
echo "== aligned"; b --workload cfg3_256c3_thick6z_aligned
echo "== aligned CG_FOLD=0"; UNIRES_CG_FOLD=0 b --workload cfg3_256c3_thick6z_aligned
echo "== cfg2"; b --workload cfg2_181c3_1mm
for c in 0 1 2; do echo "ch $c"; WL=cfg3_256c3_thick6z CH=$c bash tools/prof.sh tools/pmc5.py 2>&1 | grep "splat2<\|pull_conv2"; done
echo "== kernel trace cfg1"; WL=cfg1_181c1_denoise CH=0 bash tools/prof.sh tools/pmc5.py 2>&1 | grep -v "^rigid" | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --serial-channels --admm-iters 5 > $OUT/bench_prof_serial.log 2>&1
cp /tmp/kt2/k_kernel_stats.csv $OUT/bench_serial_kernel_stats.csv; head -14 $OUT/bench_serial_kernel_stats.csv | cut -c1-150
