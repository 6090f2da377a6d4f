#!/bin/bash
# One bench line per BASELINE configuration -> gpurun_out/r2/configs.jsonl
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/configs.jsonl
for wl in cfg1_181c1_denoise cfg2_181c3_1mm cfg3_256c3_thick6z cfg3_256c3_thick6z_aligned cfg3_256c3_thick6xyz cfg4_384c4_iso2 cfg4_384c4_iso2_gauss demo_181c3_thick4xyz; do
  python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 10 2>$OUT/cfg_$wl.err | grep '^{"metric"' >> $OUT/configs.jsonl
done
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r2/configs.jsonl')):
    d = json.loads(l)
    r = d['roofline']
    print('%-30s it/s %8.0f  matvec %8.1f us  frac %.3f  subj/s %.3f' % (d['config']['workload'], d['value'], r['us_per_launch'], r['frac'], d['subjects_per_sec']))
PY
