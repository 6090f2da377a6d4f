# y-update with the reference-default CG settings (tol 1e-3, 'max_gain'): median / min / max ms of 12 separately
# timed steps after 3 warm-ups, realised iteration counts.  WL = workload, UNIRES_CG_CHUNK = chunk size (0: full enqueue)
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
name = os.environ.get('WL', 'cfg3_256c3_thick6z')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[name], dev, seed=1234)
sett.cgs_tol = float(os.environ.get('TOL', '1e-3'))
if os.environ.get('STOP'):
    sett.cgs_stop = os.environ['STOP']
tmp = torch.zeros_like(y[0].dat)
for yc in y:
    yc.dat.zero_()
info = []
U._update_y(x, y, z, w, rho, tmp, sett, info=info)
iters = [int(r[0]) for r in info]
ts = []
for i in range(15):
    for yc in y:
        yc.dat.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    U._update_y(x, y, z, w, rho, tmp, sett)
    torch.cuda.synchronize()
    if i >= 3:
        ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print('%s chunk=%s stop=%s iters %s: median %.3f ms  min %.3f  max %.3f  -> %.3f ms per realised iteration' % (
    name, os.environ.get('UNIRES_CG_CHUNK', 'default'), sett.cgs_stop, iters, ts[len(ts) // 2], ts[0], ts[-1],
    ts[len(ts) // 2] / max(1, sum(iters))))
