#!/bin/bash
# Regenerates everything under profiles/r03_* (run on the GPU box through gpurun; outputs land in
# gpurun_out/r03, copy what is to be judged into profiles/).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# 1. the driver's command: one bench line (with cpu_baseline and variants)
python bench.py > $OUT/bench.log 2>&1; grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/r03_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command and of the explicitly serial form (at 256^3 the
#    default IS one channel after the other since settings.channel_streams = 'auto'); + one bench line
#    with the three channel streams forced
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --admm-iters 5 > $OUT/bench_prof_default.log 2>&1
cp /tmp/kt1/k_kernel_stats.csv $OUT/r03_bench_kernel_stats.csv
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --serial-channels --admm-iters 5 > $OUT/bench_prof_serial.log 2>&1
cp /tmp/kt2/k_kernel_stats.csv $OUT/r03_bench_serial_kernel_stats.csv
grep '^{"metric"' $OUT/bench_prof_serial.log | tail -1 > $OUT/r03_bench_serial.json
python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --channel-streams --admm-iters 5 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/r03_bench_streams.json
cd $GRAFT_REPO_ROOT
# 3. PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the matvec kernels: config 3 per channel,
#    configs 4 (rect and Gaussian in-plane profile), 2, 1 and the aligned / translated variants
for c in 0 1 2; do CH=$c WL=cfg3_256c3_thick6z bash tools/traffic2.sh cfg3_ch$c -- python $GRAFT_REPO_ROOT/tools/pmc5.py; done > $OUT/r03_traffic_pmc.jsonl 2>$OUT/traffic.err
python tools/traffic_summary.py $OUT/r03_traffic_pmc.jsonl $OUT/r03_traffic.json > /dev/null
for wl in cfg4_384c4_iso2 cfg4_384c4_iso2_gauss cfg2_181c3_1mm cfg1_181c1_denoise cfg3_256c3_thick6z_aligned cfg3_256c3_thick6z_shift; do
  CH=0 WL=$wl bash tools/traffic2.sh $wl -- python $GRAFT_REPO_ROOT/tools/pmc5.py
done > $OUT/r03_traffic_other_configs.jsonl 2>>$OUT/traffic.err
# 4. SQ counters of the config-3 matvec kernels
for c in 0 1 2; do echo "== channel $c"; CH=$c WL=cfg3_256c3_thick6z bash tools/pmc2.sh tools/pmc5.py; done > $OUT/r03_sq_counters.txt 2>&1
# 5. one bench line per configuration + per-kernel durations of one channel's matvecs
: > $OUT/r03_configs.jsonl; : > $OUT/r03_config_kernels.txt
for wl in cfg1_181c1_denoise cfg2_181c3_1mm cfg3_256c3_thick6z cfg3_256c3_thick6z_aligned cfg3_256c3_thick6z_shift cfg3_256c3_thick6xyz cfg4_384c4_iso2 cfg4_384c4_iso2_gauss demo_181c3_thick4xyz; do
  python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 10 2>$OUT/cfg_$wl.err | grep '^{"metric"' >> $OUT/r03_configs.jsonl
  echo "== $wl (channel 0, 20 matvecs; rocprofv3 --kernel-trace)" >> $OUT/r03_config_kernels.txt
  WL=$wl CH=0 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "unires::k_" | grep -v "build\|plan" >> $OUT/r03_config_kernels.txt
done
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r03/r03_configs.jsonl')):
    d = json.loads(l); r = d['roofline']
    print('%-30s it/s %8.0f  matvec %8.1f us  frac %.3f  subj/s %.3f' % (d['config']['workload'], d['value'], r['us_per_launch'], r['frac'], d['subjects_per_sec']))
PY
# 6. timelines of the two matvec kernels (instrumented builds; the default build is restored)
cp unires_amd/libunires_hip.so /tmp/lib_keep.so
UNIRES_HIPCC_EXTRA=-DUNIRES_S2_PROF python __graft_entry__.py --force > /tmp/prof_build.log 2>&1 || tail /tmp/prof_build.log
UNIRES_S2_PROF_OUT=/tmp/s2_timeline.txt WL=cfg3_256c3_thick6z CH=1 python tools/pmc5.py > /dev/null 2>&1
python tools/s2_timeline.py /tmp/s2_timeline.txt > $OUT/r03_splat2_timeline.txt 2>&1
UNIRES_HIPCC_EXTRA=-DUNIRES_P2_PROF python __graft_entry__.py --force > /tmp/prof_build.log 2>&1 || tail /tmp/prof_build.log
UNIRES_P2_PROF_OUT=/tmp/p2_timeline.txt WL=cfg3_256c3_thick6z CH=1 python tools/pmc5.py > /dev/null 2>&1
python tools/p2_timeline.py /tmp/p2_timeline.txt > $OUT/r03_pull2_timeline.txt 2>&1
cp /tmp/lib_keep.so unires_amd/libunires_hip.so
ls -la $OUT
