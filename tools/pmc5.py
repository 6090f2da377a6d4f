# matvec only, aligned (identity-rigid) config 3, channel 0: for PMC / kernel-trace passes
import sys, math, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
from unires_amd._project import _channel_plan
dev=torch.device('cuda:0')
wl=bench.WORKLOADS[os.environ.get('WL','cfg3_256c3_thick6z_aligned')]
x,y,z,w,rho,sett=bench.build_subject(wl,dev,seed=1234)
ch=int(os.environ.get("CH","0"))
plan=_channel_plan(x[ch],y[ch],sett.method,sett.do_proj)
print("rigid", x[ch][0].po.rigid)
p=torch.rand(y[ch].dim,device=dev); q=torch.empty_like(p)
for _ in range(int(os.environ.get("NMV", "20"))): plan.matvec(p,rho,y[ch].lam,out=q)
torch.cuda.synchronize()
