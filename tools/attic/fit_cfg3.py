# full-size exercise of the driver: a few fit() iterations of config 3 with the slice-scaling and
# rigid Gauss-Newton updates switched on; prints the time of each kind of update
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[os.environ.get('WL', 'cfg3_256c3_thick6z')], dev, seed=1234)
y = U._init_y_dat(x, y, sett)  # trilinear reslice of the observations, as the reference starts
for yc in y:
    yc.lam0 = float(yc.lam) / 4.0
sett.cgs_tol, sett.cache_atx = 1e-3, True
sett.max_iter, sett.tolerance = int(os.environ.get('ITERS', '4')), 1e-4
sett.scaling, sett.unified_rigid, sett.rigid_samp = True, True, 1
def timed(f, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3
_, t = timed(U._update_scaling, x, y, sett, max_niter_gn=1, num_linesearch=6)
print('scaling GN, 3 channels: %.1f ms' % t)
_, t = timed(U._update_rigid, x, y, sett, mean_correct=False, max_niter_gn=1, num_linesearch=6, samp=1)
print('rigid GN, 3 channels: %.1f ms' % t)
(dat, mat, R, info), t = timed(U.fit, x, y, sett)
print('fit(), %d iterations with both updates: %.1f ms; obj %s' % (info['n_iter'], t, info['obj'][:, 0].tolist()))
print('memory allocated by torch: %.2f GB' % (torch.cuda.max_memory_allocated() / 1e9))
