#!/bin/bash
# effective engine clock and VALU counters of the two general matvec kernels (config 3, channel 1)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc && WL=cfg3_256c3_thick6z CH=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pc -o p -- python $GRAFT_REPO_ROOT/tools/pmc5.py > /tmp/pc.log 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/pc/p_counter_collection.csv')))
tr = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open('/tmp/pc/p_kernel_trace.csv'))} if True else {}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name'][:40]
    if 'k_pull_conv2' in n or 'k_splat2<' in n:
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Dispatch_Id'] in tr: agg[n]['us'].append(tr[r['Dispatch_Id']])
for n, d in agg.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    print(n)
    for k, v in sorted(m.items()): print('   %-22s %.4g' % (k, v))
    if 'GRBM_GUI_ACTIVE' in m and 'us' in m:
        print('   effective clock %.2f GHz (GRBM_GUI_ACTIVE / duration under the profiler)' % (m['GRBM_GUI_ACTIVE'] / m['us'] / 1e3))
PY
