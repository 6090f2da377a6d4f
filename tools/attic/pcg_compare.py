# CG vs Jacobi-PCG vs FFT-PCG on config 3 (channel 1): relative residual after k iterations, time
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unires_amd._project import _channel_plan
dev = torch.device('cuda:0')
wl = bench.WORKLOADS[os.environ.get('WL', 'cfg3_256c3_thick6z')]
x, y, z, w, rho, sett = bench.build_subject(wl, dev, seed=1234)
c = int(os.environ.get('CH', '1'))
plan = _channel_plan(x[c], y[c], sett.method, sett.do_proj)
lam = float(y[c].lam)
for scale in (1.0, 8.0):  # 8x: the first stage of the coarse-to-fine schedule (lam x 8, rho / 8)
    l, r = lam * scale, rho / scale
    b = plan.rhs([xn.dat for xn in x[c]], w[c], z[c], r, l)
    nb = b.norm().item()
    for mode in ('none', 'jacobi', 'fft'):
        if mode != 'none':
            plan.precond_build(r, l, mode=mode)
        row = []
        for k in (2, 5, 10, 20):
            xs = torch.zeros_like(b)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            plan.cg(b, xs, r, l, max_iter=k, tolerance=0.0, precond=mode, sync=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res = (b - plan.matvec(xs, r, l)).norm().item() / nb
            row.append('k=%2d res %.2e (%.2f ms)' % (k, res, dt * 1e3))
        print('lam x%g  %-6s ' % (scale, mode), ' | '.join(row))
