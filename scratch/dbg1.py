import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import *
from tests.test_gpu_path import CASES
from oracle import nitorch_restated as N, unires_restated as O
import unires_amd as U
dev='cuda:0'
for case in ['sr_gauss_tri','sr_z_1ch']:
    prob = make_problem(seed=13, **CASES[case])
    xo,yo = oracle_structs(prob); xg,yg,sett = gpu_structs(prob,dev)
    rho=torch.tensor(prob['rho']); vx=N.voxel_size(prob['mat_y']).float()
    p=torch.rand(prob['dim_y'])*100
    ref=O.proj('AtA',p,xo[0],yo[0],method=prob['method'],do=True,rho=rho,vx_y=vx)
    out=U._proj('AtA',p.to(dev),xg[0],yg[0],method=prob['method'],do=True,rho=rho,vx_y=vx).cpu()
    print(case,'matvec err',rel_err(out,ref), 'taps', [k.tolist() for k in xg[0][0].po.smo_ker_1d], xo[0][0].po.dim_yx, xo[0][0].po.ratio)
    for it in (1,2,4,6,10,14,20):
        yr,_=run_oracle_update_y(prob,max_iter=it,tol=0.0)
        yg_,_=run_gpu_update_y(prob,dev,max_iter=it,tol=0.0)
        print('  it',it,rel_err(yg_[0].cpu(),yr[0]))
