import sys, math, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
from unires_amd._project import _channel_plan
dev=torch.device('cuda:0')
wl=bench.WORKLOADS['cfg3_256c3_thick6xyz']
x,y,z,w,rho,sett=bench.build_subject(wl,dev,seed=1234)
for c in range(3):
    plan=_channel_plan(x[c],y[c],sett.method,sett.do_proj)
    p=torch.rand(y[c].dim,device=dev); q=torch.empty_like(p)
    for _ in range(5): plan.matvec(p,rho,y[c].lam,out=q)
torch.cuda.synchronize()
