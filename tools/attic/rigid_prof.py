# one rigid Gauss-Newton update of config 3 (three channels) for rocprofv3 --kernel-trace
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[os.environ.get('WL', 'cfg3_256c3_thick6z')], dev, seed=1234)
y = U._init_y_dat(x, y, sett)
for _ in range(2):
    U._update_rigid(x, y, sett, mean_correct=False, max_niter_gn=1, num_linesearch=6, samp=1)
torch.cuda.synchronize()
