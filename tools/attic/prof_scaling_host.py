# Where the scaling Gauss-Newton step spends its time on the host (cProfile over the second and third call).
#   WL=cfg3_256c3_thick6z python tools/prof_scaling_host.py
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[os.environ.get('WL', 'cfg3_256c3_thick6z')], dev, seed=1234)
y = U._init_y_dat(x, y, sett)
sett.scaling = True
U._update_scaling(x, y, sett, max_niter_gn=1, num_linesearch=6)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(2):
    U._update_scaling(x, y, sett, max_niter_gn=1, num_linesearch=6)
    torch.cuda.synchronize()
pr.disable()
print('ms per call', (time.perf_counter() - t0) * 1e3 / 2)
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
