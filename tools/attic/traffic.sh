#!/bin/bash
# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), no trace domains
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tr_$c -o t -- python $GRAFT_REPO_ROOT/$1 > /tmp/tr.log 2>&1
done
python - <<'PY'
import csv, collections
for c in ('FETCH_SIZE','WRITE_SIZE'):
    rows=list(csv.DictReader(open('/tmp/tr_%s/t_counter_collection.csv'%c)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'unires' in r['Kernel_Name'] and r['Counter_Name']==c: agg[r['Kernel_Name'][:50]].append(float(r['Counter_Value']))
    for n,v in agg.items(): print(c, n, 'mean %.1f KB' % (sum(v)/len(v)), 'n', len(v))
PY
