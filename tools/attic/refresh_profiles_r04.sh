#!/bin/bash
# Regenerates everything under profiles/r04_* (round 4) (run on the GPU box through gpurun; outputs land in
# gpurun_out/r04, copy what is to be judged into profiles/).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# 1. PMC traffic (first: the bench lines below carry the git blob hash of profiles/r04_traffic.json) (FETCH_SIZE / WRITE_SIZE in separate passes) of the matvec kernels: config 3 per channel,
#    configs 4 (rect and Gaussian in-plane profile), 2, 1 and the aligned / translated variants
for c in 0 1 2; do CH=$c WL=cfg3_256c3_thick6z bash tools/traffic2.sh cfg3_ch$c -- python $GRAFT_REPO_ROOT/tools/pmc5.py; done > $OUT/r04_traffic_pmc.jsonl 2>$OUT/traffic.err
python tools/traffic_summary.py $OUT/r04_traffic_pmc.jsonl $OUT/r04_traffic.json > /dev/null
cp $OUT/r04_traffic.json profiles/r04_traffic.json
for wl in cfg4_384c4_iso2 cfg4_384c4_iso2_gauss cfg2_181c3_1mm cfg1_181c1_denoise cfg3_256c3_thick6z_aligned cfg3_256c3_thick6z_shift; do
  CH=0 WL=$wl bash tools/traffic2.sh $wl -- python $GRAFT_REPO_ROOT/tools/pmc5.py
done > $OUT/r04_traffic_other_configs.jsonl 2>>$OUT/traffic.err
# 2. the driver's command: one bench line (with cpu_baseline and variants)
python bench.py > $OUT/bench.log 2>&1; grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/r04_bench.json
# 3. rocprofv3 --kernel-trace --stats of the same command and of the explicitly serial form (at 256^3 the
#    default IS one channel after the other since settings.channel_streams = 'auto'); + one bench line
#    with the three channel streams forced
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --admm-iters 5 > $OUT/bench_prof_default.log 2>&1
cp /tmp/kt1/k_kernel_stats.csv $OUT/r04_bench_kernel_stats.csv
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --serial-channels --admm-iters 5 > $OUT/bench_prof_serial.log 2>&1
cp /tmp/kt2/k_kernel_stats.csv $OUT/r04_bench_serial_kernel_stats.csv
grep '^{"metric"' $OUT/bench_prof_serial.log | tail -1 > $OUT/r04_bench_serial.json
python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --channel-streams --admm-iters 5 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/r04_bench_streams.json
cd $GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
# 4. SQ counters of the config-3 matvec kernels
for c in 0 1 2; do echo "== channel $c"; CH=$c WL=cfg3_256c3_thick6z bash tools/pmc2.sh tools/pmc5.py; done > $OUT/r04_sq_counters.txt 2>&1
# 5. one bench line per configuration + per-kernel durations of one channel's matvecs
: > $OUT/r04_configs.jsonl; : > $OUT/r04_config_kernels.txt
for wl in cfg1_181c1_denoise cfg2_181c3_1mm cfg3_256c3_thick6z cfg3_256c3_thick6z_aligned cfg3_256c3_thick6z_shift cfg3_256c3_thick6xyz cfg3_256c3_thick6_orient cfg4_384c4_iso2 cfg4_384c4_iso2_gauss demo_181c3_thick4xyz; do
  python bench.py --workload $wl --no-cpu-baseline --no-variants --admm-iters 10 2>$OUT/cfg_$wl.err | grep '^{"metric"' >> $OUT/r04_configs.jsonl
  echo "== $wl (channel 0, 20 matvecs; rocprofv3 --kernel-trace)" >> $OUT/r04_config_kernels.txt
  WL=$wl CH=0 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "unires::k_" | grep -v "build\|plan" >> $OUT/r04_config_kernels.txt
done
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r04/r04_configs.jsonl')):
    d = json.loads(l); r = d['roofline']
    print('%-30s it/s %8.0f  matvec %8.1f us  frac %.3f  subj/s %.3f' % (d['config']['workload'], d['value'], r['us_per_launch'], r['frac'], d['subjects_per_sec']))
PY
# 6. the reference-default CG mode, the orientation workload's plan lines, fit() and host time, the VALU table
bash tools/r4_maxgain.sh > $OUT/maxgain.log 2>&1
bash tools/r4_orient.sh > $OUT/orient_bench.log 2>&1
for wl in tiny_32c3_thick2 cfg3_256c3_thick6z demo_181c3_thick4xyz; do WL=$wl python tools/host_time.py 2>/dev/null | tail -1; done > $OUT/r04_host_time.jsonl
for wl in cfg3_256c3_thick6z demo_181c3_thick4xyz; do WL=$wl python tools/r4_fit.py 2>/dev/null | tail -1; done > $OUT/r04_fit.jsonl
./tools/mb_valu2 > $OUT/r04_mb_valu2.txt 2>&1
./tools/mb_stream > $OUT/r04_mb_stream.txt 2>&1
bash tools/r4_clock.sh > $OUT/r04_clock.txt 2>&1
ls -la $OUT
