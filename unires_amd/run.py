"""fit() - host-side mirror of the iteration loop of unires/run.py:24-207 on device-resident
structs (no file I/O, no plotting): coarse-to-fine regularisation schedule, ADMM iterations
until the model negative log-likelihood stops improving, optional even/odd slice-scaling
updates, clean_fov post-processing.  Everything heavy goes through libunires_hip.so."""
import torch

from . import _lib
from ._host import wait_blocking
from ._lib import check, i3
from ._ops import _ptr, _stream, on_device
from ._rigid import _update_rigid
from ._update import _admm_aux, _step_size, _update_admm, _update_scaling
from .optim import get_gain
from .spatial import _m12


def _get_sched(N, sett):
    """Coarse-to-fine scaling of the regularisation (unires/_core.py:288-307):
    reg_scl = 4, sched_num = 3 -> [32, 16, 8, 4]; a single observation -> [4]."""
    if sett.sched_num < 0 or N == 1:
        sett.sched_num = 0
    if sett.rigid_mod < 1:
        sett.rigid_mod = 1
    scl = torch.as_tensor(sett.reg_scl, dtype=torch.float32).reshape(1)
    sched = (2.0 ** torch.arange(0, 32, dtype=torch.float32)).flip(0)
    ix = int(torch.min((sched - scl).abs(), dim=0)[1])
    sett.reg_scl = torch.cat((sched[ix - sett.sched_num:ix], scl))
    return sett


@on_device
def _clean_fov(x, y):
    """Zero the voxels of y[c] outside the field of view of any of its observations
    (unires/run.py:150-164)."""
    lib = _lib.load()
    for c in range(len(x)):
        for xn in x[c]:
            M = torch.linalg.solve(y[c].mat.double().cpu(), xn.po.rigid.double().cpu()
                                   .mm(torch.as_tensor(xn.mat).double().cpu())).inverse()
            check(lib.unires_clean_fov(_ptr(y[c].dat), i3(y[c].dim), _lib.c_f32x12(*_m12(M).tolist()),
                                       i3(xn.dim), _stream()))
    return y


@on_device
def fit(x, y, sett):
    """Fit model (unires/run.py:24-207).  ``x[c][n]`` / ``y[c]`` are the _input / _output
    structs with device tensors, as left by the initialisation (``_init_y_dat``,
    ``_proj_info``); ``y[c].lam0`` is the unscaled regularisation.

    Returns (dat_y, mat_y, R, info): reconstructions stacked to (dim_y, C) float32, the
    output affine, the rigid matrices (N, 4, 4) and a dict with the objective trace
    ``obj`` (n_iter, 3), the iteration count and the regularisation schedule."""
    if len(x) != len(y):
        raise ValueError('one output per channel')
    with torch.no_grad():
        dev = y[0].dat.device
        N = sum(len(xc) for xc in x)
        sett = _get_sched(N, sett)
        cnt_scl = 0
        for c in range(len(x)):
            y[c].lam = float(sett.reg_scl[cnt_scl]) * float(y[c].lam0)
        obj = torch.zeros(max(sett.max_iter, 1), 3, dtype=torch.float64, device=dev)
        tmp = torch.zeros_like(y[0].dat)
        n_done = 0
        if sett.max_iter > 0:
            rho = _step_size(x, y, sett)
            z, w = _admm_aux(y, sett)
        cnt_scl_iter = 0
        countdown0 = countdown1 = 6
        for n_iter in range(sett.max_iter):
            y, z, w, tmp, obj = _update_admm(x, y, z, w, rho, tmp, obj, n_iter, sett)
            n_done = n_iter + 1
            # one host read-back per ADMM iteration: the convergence logic below is the
            # reference's, on the same float64 objective values
            wait_blocking(dev)  # (sleep until the iteration is through: `.cpu()` alone polls)
            gain = get_gain(obj[:n_iter + 1, 0].cpu(), monotonicity='decreasing')
            if sett.do_print >= 1:
                print('{:3d} - Convergence ({} | {} | {} | gain={:0.7f})'.format(
                    n_iter, *['{:0.1f}'.format(v) for v in obj[n_iter].tolist()], float(gain)))
            if cnt_scl >= (sett.reg_scl.numel() - 1) and cnt_scl_iter > 20 \
                    and ((abs(gain) < sett.tolerance) or (n_iter >= (sett.max_iter - 1))):
                countdown0 -= 1
                if countdown0 == 0:
                    break
            else:
                countdown0 = 6
            if sett.scaling:
                x, _ = _update_scaling(x, y, sett, max_niter_gn=1, num_linesearch=6, verbose=0)
            if sett.unified_rigid and n_iter > 0 and (n_iter % sett.rigid_mod) == 0:
                x, _ = _update_rigid(x, y, sett, mean_correct=False, max_niter_gn=1,
                                     num_linesearch=6, verbose=0, samp=sett.rigid_samp)
            if cnt_scl + 1 < len(sett.reg_scl) and cnt_scl_iter > 16 and abs(gain) < 1e-3:
                countdown1 -= 1
                if countdown1 == 0:
                    cnt_scl_iter = 0
                    cnt_scl += 1
                    for c in range(len(x)):
                        y[c].lam = float(sett.reg_scl[cnt_scl]) * float(y[c].lam0)
                    rho = _step_size(x, y, sett)
            else:
                countdown1 = 6
            cnt_scl_iter += 1
        if sett.clean_fov:
            y = _clean_fov(x, y)
        R = torch.stack([xn.po.rigid.double().cpu() for xc in x for xn in xc])
        dat_y = torch.stack([yc.dat for yc in y], dim=-1)
        info = dict(obj=obj[:n_done].cpu(), n_iter=n_done, reg_scl=sett.reg_scl.clone())
        return dat_y, y[0].mat, R, info
