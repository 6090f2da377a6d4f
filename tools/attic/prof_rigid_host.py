import sys, os, torch, cProfile, pstats
sys.path.insert(0, '/root/repo')
import bench
import unires_amd as U
dev = torch.device('cuda:0')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS['cfg3_256c3_thick6z'], dev, seed=1234)
y = U._init_y_dat(x, y, sett)
U._update_rigid(x, y, sett, mean_correct=False, max_niter_gn=1, num_linesearch=6, samp=1)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    U._update_rigid(x, y, sett, mean_correct=False, max_niter_gn=1, num_linesearch=6, samp=1)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
