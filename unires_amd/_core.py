"""Initial guess of the reconstruction - counterpart of ``_init_y_dat``
(unires/_core.py:371-399).  Everything else in the
reference's ``_core.py`` (I/O, hyper-parameter estimation, coregistration,
mean-space construction) is out of scope."""
import torch

from . import _ops
from .spatial import _m12


def _init_y_dat(x, y, sett=None):
    """y[c].dat = mean over repeats of each input trilinearly resliced into the output
    space (HIP pull with on-the-fly coordinates), clamped to the input's range."""
    dim_y = tuple(y[0].dim)
    mat_y = torch.as_tensor(y[0].mat).detach().to('cpu', torch.float64)
    for c in range(len(x)):
        dat_y, sm = None, None
        for xn in x[c]:
            mat_x = torch.as_tensor(xn.mat).detach().to('cpu', torch.float64)
            mat = torch.linalg.solve(mat_x, mat_y)                 # mat_x \ mat_y
            dat = xn.dat
            mn, mx = torch.min(dat), torch.max(dat)
            res = _ops.pull_affine(dat, _m12(mat), dim_y)
            res = torch.minimum(torch.maximum(res, mn), mx)
            dat_y = res if dat_y is None else dat_y + res
            cnt = (res > 0).to(res.dtype)
            sm = cnt if sm is None else sm + cnt
        sm = torch.where(sm == 0, torch.ones_like(sm), sm)
        y[c].dat = (dat_y / sm).contiguous()
        y[c].dim = dim_y
    return y
