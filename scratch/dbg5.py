import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unires_amd import spatial
from tests.test_gpu_ops import _affines
torch.manual_seed(0)
M=_affines()['big_rigid']; sdim,gdim=(12,10,9),(11,12,10)
val=torch.rand((1,1)+gdim)
out=spatial.grid_push(val.to('cuda:0'),M,sdim).cpu()
torch.cuda.synchronize()
