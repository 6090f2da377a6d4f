# y-updates of config 3 under the reference's stopping rule (tolerance 1e-3, 'max_gain'): for a kernel trace
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
wl = bench.WORKLOADS[os.environ.get('WL', 'cfg3_256c3_thick6z')]
x, y, z, w, rho, sett = bench.build_subject(wl, dev, seed=1234)
sett.cgs_tol, sett.cgs_stop = 1e-3, os.environ.get('STOP', 'max_gain')
if os.environ.get('CS', 'auto') == 'serial':
    sett.channel_streams = False
tmp = torch.zeros_like(y[0].dat)
n = int(os.environ.get('NUP', '6'))
import time
for i in range(n + 2):
    if i == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    for yc in y:
        yc.dat.zero_()
    U._update_y(x, y, z, w, rho, tmp, sett)
torch.cuda.synchronize()
print('ms per y-update %.3f' % ((time.perf_counter() - t0) / n * 1e3))
