# An ADMM loop that just runs (for tools/helper_stack.sh to attach to): N iterations at ONLY_TOL.
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[os.environ.get('WL', 'cfg3_256c3_thick6z')], dev, seed=1234)
tmp = torch.zeros_like(y[0].dat)
sett.tolerance = 1e-4
sett.cgs_tol = float(os.environ.get('ONLY_TOL', '0.001'))
n = int(os.environ.get('N', '400'))
obj = torch.zeros((n + 4, 3), dtype=torch.float64, device=dev)
print('LOOP', flush=True)
for it in range(n):
    U._update_admm(x, y, z, w, rho, tmp, obj, it, sett)
torch.cuda.synchronize()
print('DONE', flush=True)
