#!/bin/bash
# r3 run 2: parity suite, A/B of the 16-byte aligned kernel, the flat stencil's plain-chunk path and
# the loads-first folded CG kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
b() { python bench.py --no-cpu-baseline --no-variants --admm-iters 5 "$@" 2>>$OUT/bench.err | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s it/s %8.0f  ms/step %7.3f  matvec %7.1f us (graph %.1f eager %.1f)  frac %.3f' % (d['config']['workload'], d['value'], d['ms_per_step'], r['us_per_launch'], r['us_per_launch_graph'], r['us_per_launch_eager'], r['frac']))"; }
echo "== cfg3 serial";  b --serial-channels
echo "== cfg3 serial, CG_FOLD=0"; UNIRES_CG_FOLD=0 b --serial-channels
echo "== cfg3 default"; b
echo "== cfg3 CG_FOLD=0"; UNIRES_CG_FOLD=0 b
echo "== cfg1 flat"; b --workload cfg1_181c1_denoise
echo "== cfg1 flat CG_FOLD=0"; UNIRES_CG_FOLD=0 b --workload cfg1_181c1_denoise
echo "== aligned v4"; b --workload cfg3_256c3_thick6z_aligned
echo "== aligned dword"; UNIRES_ALIGNED_V4=0 b --workload cfg3_256c3_thick6z_aligned
echo "== kernel trace cfg1"; WL=cfg1_181c1_denoise CH=0 bash tools/prof.sh tools/pmc5.py 2>&1 | grep -v "^rigid" | tail -3
echo "== kernel trace aligned"; WL=cfg3_256c3_thick6z_aligned CH=0 bash tools/prof.sh tools/pmc5.py 2>&1 | grep -v "^rigid" | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --serial-channels --admm-iters 5 > $OUT/bench_prof_serial.log 2>&1
cp /tmp/kt2/k_kernel_stats.csv $OUT/bench_serial_kernel_stats.csv; head -8 $OUT/bench_serial_kernel_stats.csv | cut -c1-150
