#!/bin/bash
# unires_plan_set_concurrency: job throughput per cap on the persistent grids (workgroups per 16 CUs), channels on streams
# (the default since r6) -> gpurun_out/r6_share_scan.txt
out=gpurun_out/r6_share_scan.txt; : > $out
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %-22s it/s %6.0f (serial %s)  matvec %6.1f us alone, %s shared  subj/s %.3f' % ('$1', '$2', d['value'], '%.0f' % d['value_channels_serial'] if d.get('value_channels_serial') else '-', r['us_per_launch'], '%.1f' % r['us_per_launch_channels_overlapped'] if r.get('us_per_launch_channels_overlapped') else '-', d['subjects_per_sec']))" >> $out; }
for wl in cfg3_256c3_thick6z cfg3_256c3_thick6xyz; do
  for s in 64 40 32 28 24 20; do UNIRES_SHARE_S2=$s python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 3 2>/dev/null | line $wl S2=$s; done
done
for wl in cfg2_181c3_1mm dn_256c3_1mm; do
  for s in 64 56 48 40 32; do UNIRES_SHARE_F1=$s python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 3 2>/dev/null | line $wl F1=$s; done
done
for wl in demo_181c3_thick4xyz; do
  for s in 64 40 28; do UNIRES_SHARE_S2=$s python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 3 2>/dev/null | line $wl S2=$s; done
done
for wl in cfg4_384c4_iso2 cfg4_384c4_iso2_gauss cfg3_256c3_thick6_orient; do
  python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 3 2>/dev/null | line $wl default
done
cat $out
