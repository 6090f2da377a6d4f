"""Unified rigid registration - host-side mirror of unires/_update.py:198-266 (_update_rigid),
:448-538 (_rigid_match) and :541-710 (_update_rigid_channel).

One Gauss-Newton step per observation on q in se(3):  rigid = expm(sum_i q_i B_i).
The reference builds ~40 volume-sized temporaries per step (Hes (dim,6), 18 dAff volumes, 27
masked products); here a step is: pull, conv, residual, conv_transpose, grid_grad with the HIP
ops, then ONE reduction kernel (unires_rigid_sums) that returns the 6-gradient and the 21
Hessian entries.  The 4x4 algebra (expm and its derivative) is float64 host work.

Basis: the reference takes nitorch's affine_basis('SE') (unires/_core.py:317).  A Gauss-Newton
step, the Armijo line search and the mean correction are invariant under any linear change of
basis of se(3) (g' = S^T g, H' = S^T H S, q' = S^-1 q give the same algebra element), so the
rigid matrices and log-likelihoods this module produces do not depend on nitorch's ordering or
sign conventions; only the numeric values of rigid_q do.  Ours: three translations, then the
rotation generators about x, y, z.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, _ops
from ._host import light_host
from ._lib import check, i3
from ._ops import _ptr, _stream, on_device
from ._project import _apply_scaling, _proj_info, _taps
from .spatial import _m12

_F64 = torch.float64


def affine_basis(group='SE', device='cpu', dtype=_F64):
    """(6, 4, 4) generators of se(3): T_x, T_y, T_z, R_x, R_y, R_z."""
    if group != 'SE':
        raise NotImplementedError("only the rigid group 'SE' is built")
    B = torch.zeros(6, 4, 4, dtype=dtype)
    for d in range(3):
        B[d, d, 3] = 1.0
    B[3, 1, 2], B[3, 2, 1] = -1.0, 1.0
    B[4, 0, 2], B[4, 2, 0] = 1.0, -1.0
    B[5, 0, 1], B[5, 1, 0] = -1.0, 1.0
    return B


def _expm_small(M):
    """Matrix exponential of a small float64 matrix: scaling and squaring with a degree-16 Taylor
    polynomial (norm scaled below 1/4: truncation error < 1e-20)."""
    import numpy as np
    nrm = np.abs(M).sum(1).max()
    sq = max(0, int(np.ceil(np.log2(max(nrm, 1e-300) * 4.0))))
    A = M / (2.0 ** sq)
    E = np.eye(M.shape[0])
    T = np.eye(M.shape[0])
    for k in range(1, 17):
        T = T @ A / k
        E = E + T
    for _ in range(sq):
        E = E @ E
    return E


def _expm(q, basis, grad_X=False):
    """rigid = expm(sum_i q_i basis_i) [, d rigid / d q_i as (num_q, 4, 4)]  (float64, host)."""
    import numpy as np
    expm = _expm_small  # tiny matrices: plain numpy beats torch's threaded CPU ops (and scipy) here
    qn = np.asarray(torch.as_tensor(q, dtype=_F64).cpu())
    Bn = np.asarray(basis.to('cpu', _F64))
    X = np.einsum('i,ijk->jk', qn, Bn)
    if not grad_X:
        return torch.from_numpy(expm(X))
    # Frechet derivative by the block trick: expm([[X, B_i], [0, X]]) = [[e^X, dexp_X(B_i)], [0, e^X]]
    dR = np.empty((Bn.shape[0], 4, 4))
    R = None
    for i in range(Bn.shape[0]):
        M = np.zeros((8, 8))
        M[:4, :4] = X
        M[4:, 4:] = X
        M[:4, 4:] = Bn[i]
        E = expm(M)
        R, dR[i] = E[:4, :4], E[:4, 4:]
    return torch.from_numpy(R.copy()), torch.from_numpy(dR)


def _logq(rigid, basis):
    """q with expm(sum q_i B_i) = rigid (principal logarithm), for structs that carry only
    po.rigid."""
    import numpy as np
    from scipy.linalg import logm
    L = torch.from_numpy(np.real(logm(torch.as_tensor(rigid, dtype=_F64).cpu().numpy())))
    A = basis.to('cpu', _F64).reshape(basis.shape[0], -1).T  # (16, num_q)
    return torch.linalg.lstsq(A, L.reshape(-1, 1)).solution[:, 0]


def _grid_matrix(po, rigid, method):
    mat = po.mat_yx if method == 'super-resolution' else po.mat_x
    dim = tuple(po.dim_yx) if method == 'super-resolution' else tuple(po.dim_x)
    return torch.linalg.solve(po.mat_y, torch.as_tensor(rigid, dtype=_F64).cpu().mm(mat)), dim


@on_device
def _rigid_terms(dat_x, dat_y, po, tau, rigid, sett, diff=False):
    """ll and, if ``diff``, the raw ingredients of the derivatives: gr3 (dim,3) = grid_grad,
    dg (dim) = residual in grid space."""
    method = sett.method
    mat, dim = _grid_matrix(po, rigid, method)
    M = _m12(mat)
    dat_yx = _ops.pull_affine(dat_y, M, dim)
    if method == 'super-resolution':
        scl = float(po.scl)
        dat_yx = _ops.conv_down(dat_yx, _taps(po), po.ratio, scl, po.dim_thick)
    dat_yx = dat_yx.reshape(tuple(dat_x.shape))
    sse = torch.zeros((), dtype=_F64, device=dat_x.device)
    check(_lib.load().unires_masked_sse(_ptr(dat_x), _ptr(dat_yx), dat_x.numel(), _ptr(sse),
                                        _stream()))
    ll = 0.5 * float(tau) * sse
    if not diff:
        return ll, None, None
    gr3 = _ops.pull_grad_affine(dat_y, M, dim).reshape(dim + (3,))
    d = dat_yx - dat_x
    d[(dat_x == 0) | (dat_yx == 0)] = 0  # :517-519
    if method == 'super-resolution':
        d = _ops.conv_up(d, _taps(po), po.ratio)  # no slice scaling here (:524)
    return ll, gr3, d.reshape(dim).contiguous()


@on_device
def _rigid_match(dat_x, dat_y, po, tau, rigid, sett, CtC=None, diff=False, verbose=0):
    """Rigid matching term with the reference's return shapes (unires/_update.py:448-538):
    ll, gr (dim, 3) = grid_grad * residual, Hes (dim, 6) = outer(grid_grad) [* CtC]."""
    ll, gr3, d = _rigid_terms(dat_x, dat_y, po, tau, rigid, sett, diff=diff)
    if not diff:
        return ll, None, None
    Hes = torch.stack([gr3[..., 0] * gr3[..., 0], gr3[..., 1] * gr3[..., 1], gr3[..., 2] * gr3[..., 2],
                       gr3[..., 0] * gr3[..., 1], gr3[..., 0] * gr3[..., 2], gr3[..., 1] * gr3[..., 2]], -1)
    if sett.method == 'super-resolution':
        Hes = Hes * CtC[..., None]
    return ll, gr3 * d[..., None], Hes


def _ctc(po, dim, device):
    """conv_transpose(conv(1)) (unires/_update.py:603-607)."""
    ones = torch.ones(dim, dtype=torch.float32, device=device)
    return _ops.conv_up(_ops.conv_down(ones, _taps(po), po.ratio), _taps(po), po.ratio).contiguous()


@on_device
@light_host
def _update_rigid_channel(xc, yc, sett, max_niter_gn=1, num_linesearch=4, verbose=0, samp=3, c=1):
    """Updates the rigid parameters of all images of one channel (unires/_update.py:541-710)."""
    lib = _lib.load()
    dev = yc.dat.device
    method = sett.method
    basis = sett.rigid_basis
    num_q = basis.shape[0]
    sll = torch.zeros((), dtype=_F64, device=dev)
    sums = torch.empty(27, dtype=_F64, device=dev)
    iu = np.triu_indices(num_q)
    for n_x in range(len(xc)):
        xn = xc[n_x]
        if xn.rigid_q is None:
            xn.rigid_q = _logq(xn.po.rigid, basis)
        q = torch.as_tensor(xn.rigid_q, dtype=_F64).cpu().clone()
        tau = float(xn.tau)
        armijo = 1.0
        po = _proj_info(xn.po.dim_y, xn.po.mat_y, xn.po.dim_x, xn.po.mat_x, rigid=xn.po.rigid,
                        prof_ip=sett.profile_ip, prof_tp=sett.profile_tp, gap=sett.gap,
                        device=dev, scl=xn.po.scl, samp=samp)
        mat = po.mat_yx if method == 'super-resolution' else po.mat_x
        dim = tuple(po.dim_yx) if method == 'super-resolution' else tuple(po.dim_x)
        # nearest-neighbour resample of the data onto the decimated lattice (unires/_update.py:589-593:
        # grid_pull at the integer coordinates D_x u, interpolation 0) = a strided slice
        sk = getattr(po, 'sk', (1, 1, 1))
        dat_x = xn.dat[::sk[0], ::sk[1], ::sk[2]][:po.dim_x[0], :po.dim_x[1], :po.dim_x[2]].contiguous()
        dat_y = yc.dat
        CtC = _ctc(po, dim, dev) if method == 'super-resolution' else None
        ll = torch.zeros((), dtype=_F64, device=dev)
        rigid = _expm(q, basis)
        for _gn in range(max_niter_gn):
            rigid, d_rigid = _expm(q, basis, grad_X=True)
            D = torch.stack([torch.linalg.solve(po.mat_y, d_rigid[i].mm(mat)) for i in range(num_q)])
            ll, gr3, dg = _rigid_terms(dat_x, dat_y, po, tau, rigid, sett, diff=True)
            d72 = (C.c_float * 72)(*D[:, :3, :].reshape(-1).float().tolist())
            check(lib.unires_rigid_sums(_ptr(gr3), _ptr(dg), _ptr(CtC) if CtC is not None else None,
                                        i3(dim), d72, _ptr(sums), _stream()))
            s = sums.cpu().numpy()  # (6x6 host algebra in numpy: torch's CPU ops cost ~10 ms each here)
            Hes = np.zeros((num_q, num_q))
            Hes[iu] = s[num_q:]
            Hes = Hes + np.triu(Hes, 1).T
            Update = torch.from_numpy(np.linalg.solve(Hes, s[:num_q]))
            old_ll, old_q, old_rigid = ll.clone(), q.clone(), rigid.clone()
            if num_linesearch == 0:
                q = old_q - armijo * Update
                rigid = _expm(q, basis)
            else:
                for n_ls in range(num_linesearch):
                    q = old_q - armijo * Update
                    rigid = _expm(q, basis)
                    ll = _rigid_terms(dat_x, dat_y, po, tau, rigid, sett)[0]
                    if bool(ll < old_ll):
                        armijo = min(1.25 * armijo, 1.0)
                        if verbose >= 1:
                            print('c={}, n={}, ls={} | :) ll={:0.2f} | q={}'.format(
                                c, n_x, n_ls, float(ll), [round(v, 7) for v in q.tolist()]))
                        break
                    ll, q, rigid = old_ll, old_q, old_rigid
                    armijo *= 0.5
        xn.rigid_q = q
        xn.po.rigid = rigid
        sll = sll + ll
    return xc, sll


@on_device
@light_host
def _update_rigid(x, y, sett, mean_correct=True, max_niter_gn=1, num_linesearch=4, verbose=0, samp=3):
    """Updates each input image's registration parameters x[c][n].rigid_q by Gauss-Newton and
    refreshes x[c][n].po.rigid (unires/_update.py:198-266).  Returns (x, sll)."""
    if getattr(sett, 'rigid_basis', None) is None:
        sett.rigid_basis = affine_basis('SE')
    dev = y[0].dat.device
    sll = torch.zeros((), dtype=_F64, device=dev)
    for c in range(len(x)):
        x[c], sllc = _update_rigid_channel(x[c], y[c], sett, max_niter_gn=max_niter_gn,
                                           num_linesearch=num_linesearch, verbose=verbose,
                                           samp=samp, c=c)
        sll = sll + sllc
    if mean_correct:
        qs = [xn.rigid_q for xc in x for xn in xc]
        mean_q = torch.stack(qs).sum(0) / float(len(qs))
        for xc in x:
            for xn in xc:
                xn.rigid_q = xn.rigid_q - mean_q
                xn.po.rigid = _expm(xn.rigid_q, sett.rigid_basis)
    return x, sll
