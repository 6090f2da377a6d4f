# What one ADMM iteration (y-update + objective + z / w updates) asks of the HOST, next to its wall (= device) time,
# for fixed-iteration CG and for the reference-default stopping rule.  With one process per GPU, eight ranks share
# the host's cores: the host share says whether they can become launch-bound.
#   wall_ms_per_iteration         : 20 iterations between device synchronisations
#   host_enqueue_ms_per_iteration : until the last launch call returned (on a busy device this includes waiting for
#                                   room in the hardware queue; with a tolerance, the chunk feeding loop)
#   host_cpu_ms_per_iteration     : process CPU time, all threads (the loop closes as run.py does: a sleeping wait
#                                   in front of the synchronisation; tools/host_profile.py splits it by thread)
# The work the host HAS to do is the same launch sequence whatever the volume: WL=tiny_32c3_thick2 makes the device
# faster than the host, so that its wall time per iteration IS the host's (the launch-bound floor).
#   WL=cfg3_256c3_thick6z python tools/host_time.py
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
from unires_amd._host import wait_blocking
dev = torch.device('cuda:0')
host = None
if os.environ.get('LOCAL_WORLD_SIZE'):  # one of several ranks on this host (tools/host_contention.py): take a share of it
    from unires_amd._host import configure_host
    host = configure_host(int(os.environ.get('LOCAL_RANK', '0')), int(os.environ['LOCAL_WORLD_SIZE']),
                          gpu_numa=os.environ.get('GPU_NUMA', '1') != '0')
name = os.environ.get('WL', 'cfg3_256c3_thick6z')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[name], dev, seed=1234)
tmp = torch.zeros_like(y[0].dat)
sett.tolerance = 1e-4
n = int(os.environ.get('N', '40'))
out = {'workload': name, 'host_cores': os.cpu_count(), 'host_config': host, 'host_pace': int(os.environ.get('PACE', '2'))}
sett.host_pace = int(os.environ.get('PACE', '2'))
for tol in (0.0, 1e-3):
    sett.cgs_tol = tol
    obj = torch.zeros((n + 4, 3), dtype=torch.float64, device=dev)
    for yc in y:
        yc.dat.zero_()
    z.zero_(), w.zero_()
    for it in range(3):
        U._update_admm(x, y, z, w, rho, tmp, obj, it, sett)
    torch.cuda.synchronize()
    c0, t0 = time.process_time(), time.perf_counter()
    for it in range(3, 3 + n):
        U._update_admm(x, y, z, w, rho, tmp, obj, it, sett)
    t_enq = time.perf_counter() - t0
    wait_blocking(dev)
    torch.cuda.synchronize()
    wall, cpu = time.perf_counter() - t0, time.process_time() - c0
    out['tol=%g' % tol] = {'wall_ms_per_iteration': wall / n * 1e3, 'host_cpu_ms_per_iteration': cpu / n * 1e3,
                           'host_enqueue_ms_per_iteration': t_enq / n * 1e3, 'host_share': cpu / wall}
print(json.dumps(out))
