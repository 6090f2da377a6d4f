#!/bin/bash
# Round 4, item 1: orientation-general operators.  Bench lines of the multi-orientation subject (with the
# plan's relabelling, and with it switched off = the r3 behaviour), of the axis-aligned subject with the
# same thick axes, and the plan-statistics lines that show which kernels serve each operator.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for wl in cfg3_256c3_thick6_orient cfg3_256c3_thick6xyz; do
  UNIRES_PULL2_VERBOSE=1 UNIRES_SPLAT2_VERBOSE=1 python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 10 \
    2>$OUT/plan_$wl.err | grep '^{"metric"' > $OUT/bench_$wl.json
  grep '^\[pull2\]\|^\[splat2\]' $OUT/plan_$wl.err | sort | uniq -c > $OUT/r04_plan_lines_$wl.txt
done
UNIRES_NO_CANON=1 python bench.py --workload cfg3_256c3_thick6_orient --no-cpu-baseline --no-variants --admm-iters 10 \
  2>/dev/null | grep '^{"metric"' > $OUT/bench_cfg3_256c3_thick6_orient_nocanon.json
python - <<'PY'
import json, os
d = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r04')
for n in ('cfg3_256c3_thick6_orient', 'cfg3_256c3_thick6xyz', 'cfg3_256c3_thick6_orient_nocanon'):
    try:
        r = json.loads(open(os.path.join(d, 'bench_%s.json' % n)).read())
        print('%-40s it/s %8.0f  matvec %8.1f us in-solve %8.1f cold  frac %.3f  subj/s %.3f' % (
            n, r['value'], r['roofline']['us_per_launch'], r['roofline']['us_per_launch_cold'], r['roofline']['frac'], r['subjects_per_sec']))
    except Exception as e:
        print(n, 'failed', e)
PY
