#!/bin/bash
# usage: tools/pmc.sh <script.py> -> per-kernel PMC averages
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/$1 > /tmp/pm.log 2>&1
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('/tmp/pm/p_counter_collection.csv')))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r['Kernel_Name'][:44]
    if 'unires' in n: agg[n][r['Counter_Name']].append(float(r['Counter_Value'])); agg[n]['VGPR'].append(float(r['VGPR_Count']))
for n,d in agg.items():
    print(n, {k.replace('SQ_',''): '%.3g'%(sum(v)/len(v)) for k,v in d.items()})
PY
