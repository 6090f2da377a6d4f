#!/bin/bash
# usage: tools/traffic2.sh <label> -- <command...> : per-kernel FETCH_SIZE / WRITE_SIZE (separate PMC passes,
# --kernel-trace only), mean per launch, written as JSON lines to stdout
label=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tr_$c -o t -- "$@" > /tmp/tr_$c.log 2>&1
done
python - "$label" <<'PY'
import csv, collections, json, sys
out = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = list(csv.DictReader(open('/tmp/tr_%s/t_counter_collection.csv' % c)))
    agg = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == c:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    for n, v in agg.items():
        out[n][c] = sum(v) / len(v)
        out[n]['launches'] = len(v)
for n, d in out.items():
    print(json.dumps({'label': sys.argv[1], 'kernel': n[:90], **d}))
PY
