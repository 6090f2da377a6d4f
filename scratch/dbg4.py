import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nitorch_restated as N
from unires_amd import spatial
from tests.test_gpu_ops import _affines
dev='cuda:0'
for name in ['big_rigid','small_rigid']:
  for sdim,gdim in [((12,10,9),(11,12,10)), ((5,70,131),(6,66,140)), ((37,29,95),(40,33,101))]:
    torch.manual_seed(0)
    M=_affines()[name]
    val=torch.rand((1,1)+gdim)
    g=N.affine_grid(M.float(),gdim)[None]
    ref=N.grid_push(val,g,sdim)
    out=spatial.grid_push(val.to(dev),M,sdim).cpu()
    d=(out-ref).abs()[0,0]
    bad=(d>1e-4).nonzero()
    print(name,sdim,gdim,'maxerr %.4f'%d.max().item(),'nbad',len(bad),'first',bad[:3].tolist(), 'sum %.3f %.3f'%(out.sum().item(), ref.sum().item()))
