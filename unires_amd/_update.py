"""ADMM updates - host-side mirror of the y-update block of unires/_update.py
(:105-152), plus _step_size (:35-64) and _admm_aux (:17-32)."""
import torch

from ._project import _channel_plan
from .spatial import voxel_size


def _admm_aux(y, sett):
    """ADMM variables z and w, (C, 3, dim_y) float32 zeros."""
    dim = (len(y), 3) + tuple(y[0].dim)
    z = torch.zeros(dim, dtype=torch.float32, device=sett.device)
    w = torch.zeros(dim, dtype=torch.float32, device=sett.device)
    return z, w


def _step_size(x, y, sett, verbose=False):
    """ADMM step size rho = rho_scl * sqrt(mean tau) / mean lam (float32 arithmetic,
    as the reference computes it on 0-d float32 tensors)."""
    rho = sett.rho
    if any(xn.ct for xc in x for xn in xc):
        rho = 1.0
    if rho is not None:
        return torch.tensor(rho, dtype=torch.float32)
    all_lam = torch.tensor([float(yc.lam) for yc in y], dtype=torch.float32)
    all_tau = torch.tensor([float(xn.tau) for xc in x for xn in xc], dtype=torch.float32)
    return sett.rho_scl * torch.sqrt(torch.mean(all_tau)) / torch.mean(all_lam)


def _update_y(x, y, z, w, rho, tmp, sett, info=None):
    """UPDATE: y  (unires/_update.py:118-152).  Per channel: assemble
    b = sum_n tau_n At x_n - lam Dt(w - rho z) into ``tmp`` and solve
    (sum tau AtA + rho lam^2 DtD) y = b by CG, in place on y[c].dat."""
    vx_y = voxel_size(y[0].mat).float()
    rho = float(rho)
    sync = info is not None
    for c in range(len(x)):
        plan = _channel_plan(x[c], y[c], sett.method, sett.do_proj, vx_y)
        lam = float(y[c].lam)
        plan.rhs([xn.dat for xn in x[c]], w[c], z[c], rho, lam, out=tmp)
        res = plan.cg(tmp, y[c].dat, rho, lam, max_iter=sett.cgs_max_iter,
                      tolerance=sett.cgs_tol, stop=sett.cgs_stop, sync=sync)
        if sync:
            info.append(res)
    return y


def _update_admm(x, y, z, w, rho, tmp, obj, n_iter, sett, info=None):
    """One ADMM iteration.  This round builds the y-update (the hot path);
    the z/w updates and the objective (unires/_update.py:154-195) are the next
    rows of SURVEY.md 8(f)."""
    y = _update_y(x, y, z, w, rho, tmp, sett, info)
    return y, z, w, tmp, obj
