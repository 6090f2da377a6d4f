# random-geometry fuzz of the push / pull kernels against the CPU oracle (run on the GPU box)
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nitorch_restated as N
from unires_amd import spatial
from tests.helpers import rigid_matrix
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(int(os.environ.get('SEED', '0')))
worst = 0.0
for it in range(int(os.environ.get('N', '40'))):
    sdim = tuple(int(v) for v in torch.randint(5, 70, (3,), generator=g))
    gdim = tuple(int(v) for v in torch.randint(5, 80, (3,), generator=g))
    u = torch.rand(9, generator=g) * 2 - 1
    M = rigid_matrix((u[:3] * 6).tolist(), (u[3:6] * 0.35).tolist())
    M[:3, :3] = M[:3, :3] @ torch.diag(1.0 + 0.5 * u[6:9].double())
    val = torch.rand((1, 1) + gdim, generator=g)
    src = torch.rand((1, 1) + sdim, generator=g)
    grid = N.affine_grid(M.float(), gdim)[None]
    ref_push = N.grid_push(val, grid, sdim)
    ref_pull = N.grid_pull(src, grid)
    out_push = spatial.grid_push(val.to(dev), M, sdim).cpu()
    out_pull = spatial.grid_pull(src.to(dev), M, gdim).cpu()
    e1 = (out_push - ref_push).abs().max().item() / max(1.0, ref_push.abs().max().item())
    e2 = (out_pull - ref_pull).abs().max().item()
    worst = max(worst, e1, e2)
    if e1 > 5e-5 or e2 > 5e-5:
        print('MISMATCH', it, sdim, gdim, M.tolist(), e1, e2)
print('fuzz done, worst error %.2e' % worst)
