#!/bin/bash
# dynamic VALU / time share of k_splat phases via the UNIRES_DBG ablation bits
cd $GRAFT_REPO_ROOT && UNIRES_HIPCC_EXTRA=-DUNIRES_ABLATE python __graft_entry__.py --force > /dev/null 2>&1  # ablation build
cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 2 4 6 8 10 18}; do
  rm -rf /tmp/pm && UNIRES_DBG=$d WL=cfg3_256c3_thick6z CH=${CH:-0} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/pmc5.py > /tmp/pm.log 2>&1
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open('/tmp/pm/p_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if 'k_splat' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
kt=list(csv.DictReader(open('/tmp/pm/p_kernel_trace.csv')))
d=[(float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3 for r in kt if 'k_splat' in r['Kernel_Name']]
d.sort()
print('dbg $d', {k.replace('SQ_INSTS_',''): '%.3g'%(sum(v)/len(v)) for k,v in agg.items()}, 'median us %.1f' % d[len(d)//2])
PY
done
cd $GRAFT_REPO_ROOT && python __graft_entry__.py --force > /dev/null 2>&1  # back to the product build
