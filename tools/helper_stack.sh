#!/bin/bash
# Which runtime thread burns CPU while a tolerance-stopped ADMM loop runs, and where: rocgdb backtraces of the
# busiest threads, taken mid-run -> gpurun_out/helper_stack.txt
mkdir -p gpurun_out
out=gpurun_out/helper_stack.txt
: > $out
N=${N:-1500} ONLY_TOL=${ONLY_TOL:-0.001} python tools/host_loop.py > gpurun_out/host_loop.log 2>&1 &
PID=$!
# wait for the loop to start
for i in $(seq 1 120); do grep -q LOOP gpurun_out/host_loop.log 2>/dev/null && break; sleep 0.5; done
sleep 2
for rep in 1 2 3; do
  echo "== sample $rep" >> $out
  # busiest threads by CPU ticks
  for t in /proc/$PID/task/*; do
    awk -v tid=$(basename $t) '{print $14+$15, tid}' $t/stat 2>/dev/null
  done | sort -rn | head -3 >> $out
  timeout 60 rocgdb -p $PID -batch -ex "thread apply all bt 14" 2>/dev/null | grep -v "^\[New\|^Reading\|^Loaded" > gpurun_out/helper_stack_raw_$rep.txt
  sleep 1
done
kill $PID 2>/dev/null
wait $PID 2>/dev/null
python - <<'PY' >> gpurun_out/helper_stack.txt
import re
for rep in (1, 2, 3):
    raw = open('gpurun_out/helper_stack_raw_%d.txt' % rep).read()
    blocks = re.split(r'\n(?=Thread \d+ )', raw)
    print('== stacks, sample', rep)
    for b in blocks:
        if 'Thread' not in b:
            continue
        # idle pool threads: skip
        if re.search(r'pthread_cond_wait|futex_wait|epoll_wait|poll \(|select \(', b) and 'hsa' not in b.lower() and 'amd' not in b.lower():
            continue
        print(b[:2500])
PY
tail -c 7000 gpurun_out/helper_stack.txt
