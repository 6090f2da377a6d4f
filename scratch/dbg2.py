import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nitorch_restated as N
from unires_amd import spatial
from tests.test_gpu_ops import _affines
torch.manual_seed(0)
dev='cuda:0'
for name in ['identity','int_shift','small_rigid']:
    for sdim,gdim in [((12,10,9),(11,12,10)), ((5,70,131),(6,66,140))]:
        M=_affines()[name]
        val=torch.rand((1,1)+gdim)
        g=N.affine_grid(M.float(),gdim)[None]
        ref=N.grid_push(val,g,sdim)
        out=spatial.grid_push(val.to(dev),M,sdim).cpu()
        d=(out-ref).abs()[0,0]
        bad=(d>1e-4).nonzero()
        print(name,sdim,gdim,'maxerr',d.max().item(),'nbad',len(bad),'first',bad[:4].tolist(), 'sum out/ref', out.sum().item(), ref.sum().item())
M=_affines()['identity']; sdim,gdim=(5,70,131),(6,66,140)
val=torch.rand((1,1)+gdim); g=N.affine_grid(M.float(),gdim)[None]
ref=N.grid_push(val,g,sdim)[0,0]; out=spatial.grid_push(val.to(dev),M,sdim).cpu()[0,0]
print('row x0 y0 out', [round(v,3) for v in out[0,0,:40].tolist()])
print('row x0 y0 ref', [round(v,3) for v in ref[0,0,:40].tolist()])
print('row x0 y1 out', [round(v,3) for v in out[0,1,:40].tolist()])
print('row x0 y1 ref', [round(v,3) for v in ref[0,1,:40].tolist()])
