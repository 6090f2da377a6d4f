# matvec only, aligned (identity-rigid) config 3, channel 0: for PMC / kernel-trace passes
import sys, math, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
from unires_amd._project import _channel_plan
dev=torch.device('cuda:0')
wl=bench.WORKLOADS[os.environ.get('WL','cfg3_256c3_thick6z_aligned')]
x,y,z,w,rho,sett=bench.build_subject(wl,dev,seed=1234)
plan=_channel_plan(x[0],y[0],sett.method,sett.do_proj)
p=torch.rand(y[0].dim,device=dev); q=torch.empty_like(p)
for _ in range(20): plan.matvec(p,rho,y[0].lam,out=q)
torch.cuda.synchronize()
