#!/bin/bash
# Regenerates everything under profiles/r06_* (round 6) (run on the GPU box through gpurun; outputs land in
# gpurun_out/r06, copy what is to be judged into profiles/).  STEPS="1 2 ..." selects parts.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
STEPS=${STEPS:-"1 2 3 4 5 7 8"}
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has 1; then
# 1. PMC traffic (first: the bench lines below carry the git blob hash of profiles/r06_traffic.json): FETCH_SIZE /
#    WRITE_SIZE in separate passes of the matvec kernels: config 3 per channel, then the other configurations -
#    config 2 now on the single-pass kernel (k_ata1), and on the pair it replaces (UNIRES_NO_ATA1=1)
for c in 0 1 2; do CH=$c WL=cfg3_256c3_thick6z bash tools/traffic2.sh cfg3_ch$c -- python $GRAFT_REPO_ROOT/tools/pmc5.py; done > $OUT/r06_traffic_pmc.jsonl 2>$OUT/traffic.err
python tools/traffic_summary.py $OUT/r06_traffic_pmc.jsonl $OUT/r06_traffic.json > /dev/null
cp $OUT/r06_traffic.json profiles/r06_traffic.json
for wl in cfg2_181c3_1mm dn_256c3_1mm cfg4_384c4_iso2 cfg4_384c4_iso2_gauss cfg1_181c1_denoise cfg3_256c3_thick6z_aligned cfg3_256c3_thick6z_shift; do
  CH=1 WL=$wl bash tools/traffic2.sh $wl -- python $GRAFT_REPO_ROOT/tools/pmc5.py
done > $OUT/r06_traffic_other_configs.jsonl 2>>$OUT/traffic.err
for wl in cfg2_181c3_1mm dn_256c3_1mm; do
  UNIRES_NO_ATA1=1 CH=1 WL=$wl bash tools/traffic2.sh ${wl}_pair -- python $GRAFT_REPO_ROOT/tools/pmc5.py
done >> $OUT/r06_traffic_other_configs.jsonl 2>>$OUT/traffic.err
fi
if has 2; then
# 2. the driver's command: one bench line (with cpu_baseline and variants)
python bench.py > $OUT/bench.log 2>&1; grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/r06_bench.json
fi
if has 3; then
# 3. rocprofv3 --kernel-trace --stats of the same command and of the explicitly serial form; + one bench line
#    with the three channel streams forced
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --admm-iters 5 > $OUT/bench_prof_default.log 2>&1
cp /tmp/kt1/k_kernel_stats.csv $OUT/r06_bench_kernel_stats.csv
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --serial-channels --admm-iters 5 > $OUT/bench_prof_serial.log 2>&1
cp /tmp/kt2/k_kernel_stats.csv $OUT/r06_bench_serial_kernel_stats.csv
grep '^{"metric"' $OUT/bench_prof_serial.log | tail -1 > $OUT/r06_bench_serial.json
# ... and the same without the leg under the reference's stopping rule: every matvec launch of this run is a real one (the
# guarded rule's launches that return at entry deflate the averages above: k_splat2 61 us over 4 233 calls)
rm -rf /tmp/kt2b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2b -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --no-tol-leg --serial-channels --admm-iters 5 > $OUT/bench_prof_serial_fixed.log 2>&1
cp /tmp/kt2b/k_kernel_stats.csv $OUT/r06_bench_serial_fixed_kernel_stats.csv
grep '^{"metric"' $OUT/bench_prof_serial_fixed.log | tail -1 > $OUT/r06_bench_serial_fixed.json
python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --channel-streams --admm-iters 5 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/r06_bench_streams.json
# ... and of config 2 (the single-pass kernel inside the CG solves)
rm -rf /tmp/kt3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o k -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2_181c3_1mm --no-cpu-baseline --no-variants --serial-channels --admm-iters 5 > $OUT/bench_prof_cfg2.log 2>&1
cp /tmp/kt3/k_kernel_stats.csv $OUT/r06_bench_cfg2_serial_kernel_stats.csv
cd $GRAFT_REPO_ROOT
fi
if has 4; then
# 4. SQ counters: the config-3 matvec kernels, and the single-pass kernel on config 2 and at 256^3
for c in 0 1 2; do echo "== channel $c"; CH=$c WL=cfg3_256c3_thick6z bash tools/pmc2.sh tools/pmc5.py; done > $OUT/r06_sq_counters.txt 2>&1
for wl in cfg2_181c3_1mm dn_256c3_1mm; do for c in 0 1; do echo "== $wl channel $c"; CH=$c WL=$wl bash tools/pmc2.sh tools/pmc5.py | grep -A22 "k_ata1"; done; done > $OUT/r06_ata1_sq_counters.txt 2>&1
bash tools/r4_clock.sh > $OUT/r06_clock.txt 2>&1
fi
if has 5; then
# 5. one bench line per configuration + per-kernel durations of one channel's matvecs
: > $OUT/r06_configs.jsonl; : > $OUT/r06_config_kernels.txt
for wl in cfg1_181c1_denoise cfg2_181c3_1mm dn_256c3_1mm cfg3_256c3_thick6z cfg3_256c3_thick6z_aligned cfg3_256c3_thick6z_shift cfg3_256c3_thick6xyz cfg3_256c3_thick6_orient cfg4_384c4_iso2 cfg4_384c4_iso2_gauss demo_181c3_thick4xyz; do
  python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 10 2>$OUT/cfg_$wl.err | grep '^{"metric"' >> $OUT/r06_configs.jsonl
  for c in 0 1; do
    echo "== $wl (channel $c, 20 matvecs; rocprofv3 --kernel-trace)" >> $OUT/r06_config_kernels.txt
    WL=$wl CH=$c bash tools/prof.sh tools/pmc5.py 2>&1 | grep "unires::k_" | grep -v "build\|plan" >> $OUT/r06_config_kernels.txt
  done
done
for wl in cfg2_181c3_1mm dn_256c3_1mm; do
  UNIRES_NO_ATA1=1 python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 10 2>/dev/null | grep '^{"metric"' | sed 's/"workload": "\([a-z0-9_]*\)"/"workload": "\1 (UNIRES_NO_ATA1=1: the pull + splat pair)"/' >> $OUT/r06_configs.jsonl
  for c in 0 1; do
    echo "== $wl with UNIRES_NO_ATA1=1: the pull + splat pair of round 4 (channel $c)" >> $OUT/r06_config_kernels.txt
    UNIRES_NO_ATA1=1 WL=$wl CH=$c bash tools/prof.sh tools/pmc5.py 2>&1 | grep "unires::k_" | grep -v "build\|plan" >> $OUT/r06_config_kernels.txt
  done
done
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r06/r06_configs.jsonl')):
    d = json.loads(l); r = d['roofline']
    print('%-60s it/s %8.0f  matvec %8.1f us  frac %.3f  subj/s %.3f' % (d['config']['workload'], d['value'], r['us_per_launch'], r['frac'], d['subjects_per_sec']))
PY
fi
if has 6; then
# 6. the reference-default CG mode, the orientation workload's plan lines
bash tools/r4_maxgain.sh > $OUT/maxgain.log 2>&1
bash tools/r4_orient.sh > $OUT/orient_bench.log 2>&1
fi
if has 7; then
# 7. host side: fit(), host time with the pacer, eight ranks' host loops, how to wait without burning a core
bash tools/r5_host.sh > $OUT/host.log 2>&1
{ python tools/wait_probe.py; MODE=flags python tools/wait_probe.py; } > $OUT/r06_wait_probe.txt 2>/dev/null
# the single-pass kernel's packing variants and what the schedule builds cost
{ for v in "UNIRES_F1_PACK=0 UNIRES_F1_EXACT=0" "UNIRES_F1_PACK=1 UNIRES_F1_EXACT=0" "UNIRES_F1_PACK=1 UNIRES_F1_EXACT=1"; do
    echo "== $v"; env $v UNIRES_ATA1_VERBOSE=1 WL=cfg2_181c3_1mm python tools/f1_check.py 2>&1 | grep "ata1\] tile\|matvec"; done
  echo "== schedule build kernels (rocprofv3), thorough / quick"
  for ex in 1 0; do UNIRES_S2_EXACT=$ex UNIRES_F1_EXACT=$ex WL=cfg3_256c3_thick6z CH=1 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "build"; UNIRES_S2_EXACT=$ex UNIRES_F1_EXACT=$ex WL=cfg2_181c3_1mm CH=1 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "build"; done
} > $OUT/r06_ata1_packing.txt 2>&1
fi
if has 8; then
# 8. round 6: channels overlapped by default - throughput per cap on the persistent grids and its repeatability; phase
#    ablations of the pair at HEAD (needs build/ab/abl.so: tools/r6_buildlib.sh abl -DUNIRES_ABLATE); L2 counters; the
#    reference-default stopping rule's kernel account and the gains its guard band sees
bash tools/r6_share_scan.sh > /dev/null 2>&1; cp gpurun_out/r6_share_scan.txt $OUT/r06_share_scan.txt
{ echo "# cfg3_256c3_thick6z, three bench.py runs per cap (UNIRES_SHARE_S2 = splat workgroups per 16 CUs: 64 = no cap, 28 = 448 of 1024, the default)"
  for s in 64 28; do for rep in 1 2 3; do UNIRES_SHARE_S2=$s python bench.py --no-cpu-baseline --no-variants --admm-iters 1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('S2=$s it/s %6.0f (serial %.0f)' % (d['value'], d['value_channels_serial'] or 0))"; done; done; } > $OUT/r06_share_repeat.txt
[ -f build/ab/abl.so ] || bash tools/r6_buildlib.sh abl -DUNIRES_ABLATE > /dev/null 2>&1
bash tools/r6_ablate.sh > $OUT/r06_ablations.txt 2>&1
for wl in cfg3_256c3_thick6z cfg4_384c4_iso2; do echo "== $wl"; WL=$wl CH=1 bash tools/pmc_tcc.sh; done > $OUT/r06_l2_counters.txt 2>&1
bash tools/r6_tolprof.sh > $OUT/r06_tol_kernels.txt 2>&1
python tools/r6_gains.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_gains.txt
fi
ls -la $OUT
