#!/bin/bash
# usage: tools/pmc2.sh <script.py> -> per-kernel SQ counters, two PMC passes (8 SQ slots each)
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
rm -rf /tmp/pm1 /tmp/pm2
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d /tmp/pm1 -o p -- python $GRAFT_REPO_ROOT/$1 > /tmp/pm1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d /tmp/pm2 -o p -- python $GRAFT_REPO_ROOT/$1 > /tmp/pm2.log 2>&1
python - <<'PY'
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('/tmp/pm1','/tmp/pm2'):
    try: rows=list(csv.DictReader(open(d+'/p_counter_collection.csv')))
    except Exception as e: print('missing', d, e); continue
    for r in rows:
        n=r['Kernel_Name'][:44]
        if 'unires' in n:
            agg[n][r['Counter_Name']].append(float(r['Counter_Value'])); agg[n]['VGPR']=[float(r['VGPR_Count'])]; agg[n]['LDS']=[float(r.get('LDS_Block_Size',0) or 0)]
for n,d in agg.items():
    print(n)
    for k,v in sorted(d.items()): print('   %-24s %.4g' % (k.replace('SQ_',''), sum(v)/len(v)))
PY
