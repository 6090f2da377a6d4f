# End-to-end fit() at full size with what the reference's demos switch on (demos/demo_multi_channel.ipynb:224:
# scaling=True, unified_rigid=True): ms per ADMM iteration, per update, and the HOST's share - process CPU time
# (time.process_time: all threads of this process) next to wall time.  One JSON line per workload.
#   WL=cfg3_256c3_thick6z ITERS=12 python tools/r4_fit.py
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
name = os.environ.get('WL', 'cfg3_256c3_thick6z')
iters = int(os.environ.get('ITERS', '12'))


def fresh():
    x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[name], dev, seed=1234)
    y = U._init_y_dat(x, y, sett)  # trilinear reslice of the observations, as the reference starts
    for yc in y:
        yc.lam0 = float(yc.lam) / 4.0
    sett.cgs_tol, sett.cgs_max_iter, sett.cache_atx = 1e-3, 20, True  # the reference's defaults (struct.py:65-67)
    sett.max_iter, sett.tolerance = iters, 0.0  # (tolerance 0: all `iters` iterations run)
    sett.rigid_samp = 1
    return x, y, z, w, rho, sett


def timed(f, *a, **k):
    torch.cuda.synchronize()
    c0, t0 = time.process_time(), time.perf_counter()
    r = f(*a, **k)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3, (time.process_time() - c0) * 1e3, t_enq * 1e3


out = {'workload': name, 'admm_iterations': iters, 'cg': 'tol 1e-3, max_gain, <= 20 iterations (reference default)'}
x, y, z, w, rho, sett = fresh()
sett.scaling, sett.unified_rigid = False, False
U.fit(x, y, sett)  # warm-up: plans, graphs
x, y, z, w, rho, sett = fresh()
sett.scaling, sett.unified_rigid = False, False
(_, _, _, info), wall, cpu, enq = timed(U.fit, x, y, sett)
out['fit_plain'] = {'ms_per_admm_iteration': wall / info['n_iter'], 'host_cpu_ms_per_iteration': cpu / info['n_iter'],
                    'n_iter': int(info['n_iter'])}
x, y, z, w, rho, sett = fresh()
sett.scaling, sett.unified_rigid = True, True
# (r5: mean of 8 consecutive steps after a warm-up.  Process CPU time over ONE 5 ms step is tick-accounted: a helper
# thread that is merely awake at a 10 ms tick is charged the whole tick - r4's "32 ms of CPU for a 5 ms rigid step"
# was wall + three ticks, tools/rigid_threads.py)
NREP = 8


def many(f, *a, **k):
    for _ in range(NREP):
        f(*a, **k)


timed(U._update_scaling, x, y, sett, max_niter_gn=1, num_linesearch=6)
_, wall, cpu, enq = timed(many, U._update_scaling, x, y, sett, max_niter_gn=1, num_linesearch=6)
out['scaling_gn_all_channels'] = {'ms': wall / NREP, 'host_cpu_ms': cpu / NREP, 'steps_timed': NREP}
timed(U._update_rigid, x, y, sett, mean_correct=False, max_niter_gn=1, num_linesearch=6, samp=1)
_, wall, cpu, enq = timed(many, U._update_rigid, x, y, sett, mean_correct=False, max_niter_gn=1, num_linesearch=6, samp=1)
out['rigid_gn_all_channels'] = {'ms': wall / NREP, 'host_cpu_ms': cpu / NREP, 'steps_timed': NREP}
x, y, z, w, rho, sett = fresh()
sett.scaling, sett.unified_rigid = True, True
(_, _, _, info), wall, cpu, enq = timed(U.fit, x, y, sett)
out['fit_scaling_rigid'] = {'ms_per_admm_iteration': wall / info['n_iter'], 'host_cpu_ms_per_iteration': cpu / info['n_iter'],
                            'n_iter': int(info['n_iter'])}
out['torch_max_memory_GB'] = torch.cuda.max_memory_allocated() / 1e9
print(json.dumps(out))
