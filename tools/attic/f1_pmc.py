# a few CG matvecs per channel of a workload (PMC / kernel-trace subject): WL, CH (channel, default all)
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unires_amd._project import _channel_plan
dev = torch.device('cuda:0')
wl = bench.WORKLOADS[os.environ.get('WL', 'cfg2_181c3_1mm')]
x, y, z, w, rho, sett = bench.build_subject(wl, dev, seed=1234)
chans = [int(os.environ['CH'])] if os.environ.get('CH') else range(len(x))
for c in chans:
    pl = _channel_plan(x[c], y[c], sett.method, sett.do_proj)
    ps = [torch.rand(y[c].dim, device=dev) for _ in range(3)]
    q = torch.empty_like(ps[0])
    for i in range(9):
        pl.matvec(ps[i % 3], rho, y[c].lam, out=q)
torch.cuda.synchronize()
