"""Which kernels does the A^T A of a tests/helpers.make_problem case launch?  (development aid: run under
tools/prof.sh;  CASE='dict(dim_y=(20,24,32), thick=2, regime="sr", iso=True, prof_ip=2)' [CHN=<channel>])"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import make_problem, gpu_structs
import unires_amd as U
dev = torch.device('cuda:0')
ch = int(os.environ.get('CHN', '0'))
prob = make_problem(seed=3, n_channels=ch + 1, **eval(os.environ['CASE']))
xg, yg, sett = gpu_structs(prob, dev)
p = torch.rand(prob['dim_y'], device=dev)
q = U._proj('AtA', p, xg[ch], yg[ch], method=prob['method'], do=prob['do_proj'], rho=torch.tensor(prob['rho']),
            vx_y=torch.ones(3))
torch.cuda.synchronize()
print('taps', [t.numel() for t in xg[ch][0].po.smo_ker], 'dim_x', tuple(xg[ch][0].po.dim_x), 'dim_thick', xg[ch][0].po.dim_thick)
