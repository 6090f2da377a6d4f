#!/bin/bash
# usage: WL=... CH=... tools/pmc_tcc.sh -> per-kernel L2 (TCC) counters of the matvec kernels: requests, hits, misses,
# reads to the fabric (one PMC pass, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmt
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum --output-format csv -d /tmp/pmt -o p -- python $GRAFT_REPO_ROOT/tools/pmc5.py > /tmp/pmt.log 2>&1
python - <<'PY'
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
try: rows=list(csv.DictReader(open('/tmp/pmt/p_counter_collection.csv')))
except Exception as e: print('missing', e); print(open('/tmp/pmt.log').read()[-1500:]); rows=[]
for r in rows:
    n=r['Kernel_Name'][:60]
    if 'unires::k_' in n and 'build' not in n and 'plan' not in n:
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,d in agg.items():
    m={k:sum(v)/len(v) for k,v in d.items()}
    print('%-62s' % n, '  '.join('%s %.4g' % (k.replace('TCC_','').replace('_sum',''), v) for k,v in sorted(m.items())),
          ' hit rate %.3f' % (m.get('TCC_HIT_sum',0)/max(m.get('TCC_HIT_sum',0)+m.get('TCC_MISS_sum',0),1)))
PY
