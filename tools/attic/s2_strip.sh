#!/bin/bash
# y-strip processing order of k_splat2's tiles (UNIRES_S2_STRIP: tiles per strip, 0 = index order) and the band
# height of k_pull_conv2's workgroup walk (UNIRES_P2_BAND) -> matvec times inside the solve / cold, per workload
mkdir -p gpurun_out
out=gpurun_out/s2_strip.txt
: > $out
for wl in ${WLS:-cfg4_384c4_iso2 cfg3_256c3_thick6z}; do
  for s in ${STRIPS:-0 -1 4 8 16 32}; do
    env UNIRES_S2_STRIP=$s python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$wl strip $s: it/s %.0f  in-solve %.1f us  cold %.1f us  by channel %s' % (d['value'], r['us_per_launch'], r['us_per_launch_cold'], ' '.join('%.1f' % v for v in r['us_per_launch_by_channel'])))
" >> $out
  done
done
cat $out
