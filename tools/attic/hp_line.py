# one-line-per-case summary of tools/host_profile.py's JSON (stdin)
import json, sys
d = json.load(sys.stdin)
for k in ('tol=0', 'tol=0.001'):
    for c in ('sleep', 'spin'):
        r = d[k][c]
        print(d['workload'], k, c, 'wall %.2f process %.2f main %.2f others %.2f' % (
            r['wall_ms'], r['process_cpu_ms'], r['main_thread_cpu_ms'], r['other_threads_cpu_ms']), r['threads_cpu_ms'])
