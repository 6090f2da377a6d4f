# Eight ranks on one host, as the batch mode runs them (one process per GPU) - shown with what a 1-GPU box has:
# N copies of the ADMM host loop (tools/host_time.py) on a LAUNCH-BOUND subject (32^3: the device finishes
# every kernel before the host has enqueued the next, so an iteration's wall time is host time) started together
# against the one GPU.  If the ranks' host sides got in each other's way - oversubscribed cores, threads
# migrating, spinning waits - the wall time per iteration of 8 ranks would be far above 8 x one rank's share of
# the (shared) device; pinned, thread-capped and sleeping while they wait, they should queue on the device only.
#   python tools/host_contention.py            -> one JSON line per N in (1, 8)
import json, os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
wl = os.environ.get('WL', 'tiny_32c3_thick2')
for n in (1, 8):
    procs = []
    t0 = time.perf_counter()
    for r in range(n):
        env = dict(os.environ, WL=wl, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n), GPU_NUMA='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, 'host_time.py')], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    outs = [p.communicate()[0] for p in procs]
    recs = [json.loads(o.strip().splitlines()[-1]) for o in outs if o.strip()]
    row = {'ranks': n, 'workload': wl, 'launch_to_exit_s': time.perf_counter() - t0}
    for key in ('tol=0', 'tol=0.001'):
        w = [r[key]['wall_ms_per_iteration'] for r in recs]
        c = [r[key]['host_cpu_ms_per_iteration'] for r in recs]
        row[key] = {'wall_ms_per_iteration_mean': sum(w) / len(w), 'wall_ms_per_iteration_max': max(w),
                    'host_cpu_ms_per_iteration_mean': sum(c) / len(c)}
    row['cpus_of_rank'] = [r['host_config']['cpus'] if r.get('host_config') else None for r in recs][:8]
    row['threads'] = recs[0]['host_config']['threads'] if recs and recs[0].get('host_config') else None
    print(json.dumps(row))
