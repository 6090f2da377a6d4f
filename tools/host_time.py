import sys, os, time, torch, cProfile, pstats
sys.path.insert(0, '/root/repo')
import bench
import unires_amd as U
dev = torch.device('cuda:0')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS['cfg3_256c3_thick6z'], dev, seed=1234)
tmp = torch.zeros_like(y[0].dat)
for _ in range(3): U._update_y(x, y, z, w, rho, tmp, sett)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): U._update_y(x, y, z, w, rho, tmp, sett)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host enqueue per y-update: %.2f ms; total per y-update %.2f ms' % ((t1 - t0) * 100, (t2 - t0) * 100))
obj = torch.zeros((64, 3), dtype=torch.float64, device=dev); sett.tolerance = 1e-4
pr = cProfile.Profile(); pr.enable()
for it in range(5): U._update_admm(x, y, z, w, rho, tmp, obj, it, sett)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
