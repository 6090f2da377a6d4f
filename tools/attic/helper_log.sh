#!/bin/bash
# What the runtime's threads log (AMD_LOG_LEVEL=4) during a few tolerance-stopped ADMM iterations: message
# templates per thread -> gpurun_out/helper_log.txt
mkdir -p gpurun_out
rm -f /tmp/amdlog*
N=${N:-6} ONLY_TOL=${ONLY_TOL:-0.001} AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=/tmp/amdlog python tools/host_loop.py > gpurun_out/host_loop.log 2>&1
ls -la /tmp/amdlog* >> gpurun_out/host_loop.log
python - <<'PY' > gpurun_out/helper_log.txt
import re, glob, collections
lines = []
for f in glob.glob('/tmp/amdlog*'):
    lines += open(f, errors='replace').read().splitlines()
print('lines', len(lines))
# keep the tail: the loop itself (the set-up logs far more)
tail = lines[-60000:]
by = collections.defaultdict(collections.Counter)
pat = re.compile(r'\[pid:(\d+)\s+tid:\s*(0x[0-9a-f]+|\d+)\]\s*(.*)')
for l in tail:
    m = pat.search(l)
    if not m:
        continue
    msg = re.sub(r'0x[0-9a-fA-F]+', 'X', m.group(3))
    msg = re.sub(r'\d+', 'N', msg)[:140]
    by[m.group(2)][msg] += 1
for tid, c in sorted(by.items(), key=lambda kv: -sum(kv[1].values())):
    print('== tid', tid, 'lines', sum(c.values()))
    for msg, n in c.most_common(14):
        print('   %6d  %s' % (n, msg))
print('== helper-thread lines that are not plain completions, with the line of the main thread before each')
helper = sorted(by.items(), key=lambda kv: -sum(kv[1].values()))[1][0]
prev = ''
shown = 0
for l in tail:
    if helper in l:
        if 'complete' not in l or 'Wall' in l:
            print('   main:', prev[:200])
            print('   HELP:', l[:200])
            shown += 1
            if shown > 40:
                break
    else:
        prev = l
print('== markers / waits on the main thread')
c = collections.Counter()
for l in tail:
    if helper in l:
        continue
    if re.search(r'[Mm]arker|[Ww]ait|[Bb]arrier|signal|Signal|flush', l):
        m = pat.search(l)
        if m:
            msg = re.sub(r'0x[0-9a-fA-F]+', 'X', m.group(3)); msg = re.sub(r'\d+', 'N', msg)[:160]
            c[msg] += 1
for msg, n in c.most_common(25):
    print('   %6d  %s' % (n, msg))
PY
head -c 9000 gpurun_out/helper_log.txt
