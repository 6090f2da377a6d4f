# usage: python tools/traffic_summary.py <traffic_pmc.jsonl> <out.json>
# Mean HBM bytes per launch of the two config-3 matvec kernels from the PMC passes of tools/traffic2.sh.
# Counters are KB; FETCH_SIZE is doubled (calibration: profiles/r02_traffic_calibration.jsonl).
import collections
import json
import sys

ALG = 156237824  # B_mv of config 3, SURVEY 8(d)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open(sys.argv[1]):
    r = json.loads(line)
    k = r['kernel']
    name = 'pull_conv2' if 'k_pull_conv2' in k else ('splat2' if 'k_splat2<' in k else None)
    if name:
        acc[name]['fetch'].append(r['FETCH_SIZE'] * 1024 * 2)
        acc[name]['write'].append(r['WRITE_SIZE'] * 1024)
out = {
    'workload': 'cfg3_256c3_thick6z',
    'note': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE in separate passes (tools/traffic2.sh), '
            'counters are KB; FETCH_SIZE doubled: calibrated on known-bytes copy / read kernels at 4, 8 and 16 '
            'bytes per lane (profiles/r02_traffic_calibration.jsonl: 131 078 KB reported for 262 144 KB read at '
            'every width; WRITE_SIZE exact)',
    'per_kernel_mean_over_channels': {}}
total = 0.0
for name, v in acc.items():
    f, w = sum(v['fetch']) / len(v['fetch']), sum(v['write']) / len(v['write'])
    out['per_kernel_mean_over_channels'][name] = {'fetch_bytes': f, 'write_bytes': w}
    total += f + w
out['bytes_per_launch'], out['algorithmic_bytes'], out['ratio'] = int(total), ALG, total / ALG
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out, indent=1))
