"""Restatement (torch-CPU) of the UniRes y-update path, composed exactly as the
reference composes it (dense affine grid rebuilt per call, unfused pull / conv /
push, stacked gradient, unpreconditioned CG) - deliberately slow.

ORACLE - test infrastructure only (see oracle/__init__.py).  This half is pinned
against the reference's own Python (tests/test_reference_pin.py, fixtures written
by tests/golden/make_golden_from_reference.py); PARITY UNPINNED at the nitorch
boundary (oracle/nitorch_restated.py).  Each function cites the reference lines
it follows.
"""
import math
from types import SimpleNamespace

import torch
from torch.nn import functional as F

from .nitorch_restated import (affine_grid, grid_pull, grid_push, im_gradient,
                               im_divergence, cg, smooth, voxel_size)


# --------------------------------------------------------------------------
# unires/struct.py:4-54  (only the fields the hot path reads)
# --------------------------------------------------------------------------
def make_input(dat, mat, tau, po=None):
    return SimpleNamespace(dat=dat, dim=tuple(dat.shape), mat=mat, tau=tau, po=po, ct=False)


def make_output(dat, mat, lam):
    return SimpleNamespace(dat=dat, dim=tuple(dat.shape), mat=mat, lam=lam, lam0=lam)


# --------------------------------------------------------------------------
# unires/_project.py:9-24
# --------------------------------------------------------------------------
def apply_scaling(dat, scl, dim):
    """Even/odd slice scaling along ``dim``: even *= exp(scl), odd *= exp(-scl)."""
    scl = torch.as_tensor(scl, dtype=dat.dtype)
    out = torch.zeros_like(dat)
    sl_e = [Ellipsis, slice(None), slice(None), slice(None)]
    sl_o = [Ellipsis, slice(None), slice(None), slice(None)]
    sl_e[1 + dim] = slice(0, None, 2)
    sl_o[1 + dim] = slice(1, None, 2)
    out[tuple(sl_e)] = torch.exp(scl) * dat[tuple(sl_e)]
    out[tuple(sl_o)] = torch.exp(-scl) * dat[tuple(sl_o)]
    return out


# --------------------------------------------------------------------------
# unires/_project.py:193-297
# --------------------------------------------------------------------------
def proj_info(dim_y, mat_y, dim_x, mat_x, rigid=None, prof_ip=0, prof_tp=0, gap=0.0,
              scl=0.0, gauss_lim=None, samp=0):
    """Projection-operator descriptor; ``samp > 0``: the sub-sampling branch (:245-264) the rigid
    Gauss-Newton uses (low-res side only: the high-res branch is dead code, :255 compares vx_x
    with itself)."""
    dt = torch.float64
    po = SimpleNamespace()
    mat_y = torch.as_tensor(mat_y, dtype=dt)
    mat_x = torch.as_tensor(mat_x, dtype=dt)
    dim_y_t = torch.tensor(tuple(dim_y), dtype=dt)
    dim_x_t = torch.tensor(tuple(dim_x), dtype=dt)
    po.mat_y, po.vx_y = mat_y, voxel_size(mat_y)              # :223-224
    po.mat_x, po.vx_x = mat_x, voxel_size(mat_x)              # :229-230
    ndim = 3
    po.rigid = torch.eye(4, dtype=dt) if rigid is None else torch.as_tensor(rigid).to(dt)  # :234-237
    gap_cn = torch.zeros(ndim, dtype=dt)                       # :239
    profile = torch.tensor((prof_ip,) * ndim, dtype=dt)        # :240
    dim_thick = int(torch.max(po.vx_x, dim=0)[1])              # :241
    gap_cn[dim_thick] = gap                                    # :242
    profile[dim_thick] = prof_tp                               # :243
    po.dim_thick = dim_thick
    po.D_x = None
    if samp > 0:                                               # :245-253, :264
        one = torch.ones(ndim, dtype=dt)
        sk = torch.max(one, torch.floor(samp * one / po.vx_x + 0.5))
        po.D_x = torch.diag(torch.cat((sk, one[0, None])))
        mat_x = mat_x.mm(po.D_x)
        dim_x_t = po.D_x.inverse()[:ndim, :ndim].mm(dim_x_t[..., None]).floor().squeeze()
        po.mat_x, po.vx_x = mat_x, voxel_size(mat_x)
    ratio = torch.linalg.solve(mat_y, mat_x)                   # :266
    ratio = (ratio[:ndim, :ndim] ** 2).sum(0).sqrt()           # :267
    ratio = ratio.ceil().clamp(1)                              # :268
    mat_yx = torch.cat((ratio, torch.ones(1, dtype=dt))).diag()  # :269
    po.mat_yx = mat_x.matmul(mat_yx.inverse())                 # :270
    dim_yx = (dim_x_t - 1) * ratio + 1                         # :271
    profile[ratio == 1] = -1                                   # :273
    profile = profile.int().tolist()                           # :274
    fwhm = (1.0 - gap_cn) * ratio                              # :276
    po.smo_ker = smooth(profile, fwhm.tolist(), sep=False, dtype=torch.float32,
                        gauss_lim=gauss_lim)                   # :277
    po.smo_ker_1d = smooth(profile, fwhm.tolist(), sep=True, dtype=torch.float32,
                           gauss_lim=gauss_lim)
    off = torch.tensor(po.smo_ker.shape[-ndim:], dtype=dt)     # :280
    off = torch.div(-(off - 1), 2, rounding_mode='floor')      # :281
    mat_off = torch.eye(ndim + 1, dtype=dt)                    # :282
    mat_off[:ndim, -1] = off                                   # :283
    dim_yx = dim_yx + 2 * torch.abs(off)                       # :284
    po.mat_yx = torch.matmul(po.mat_yx, mat_off)               # :285
    po.scl = torch.as_tensor(scl, dtype=torch.float32)         # :287-290
    po.dim_y = tuple(dim_y_t.int().tolist())                   # :292
    po.dim_yx = tuple(dim_yx.int().tolist())                   # :293
    po.dim_x = tuple(dim_x_t.int().tolist())                   # :294
    po.ratio = tuple(ratio.int().tolist())                     # :295
    return po


def proj_matrix(po, method):
    """mat_y \\ (rigid @ mat_yx|mat_x), float64  (unires/_project.py:145-150)."""
    if method == 'super-resolution':
        return torch.linalg.solve(po.mat_y, po.rigid.mm(po.mat_yx)), po.dim_yx
    if method == 'denoising':
        return torch.linalg.solve(po.mat_y, po.rigid.mm(po.mat_x)), po.dim_x
    raise ValueError('Undefined method')


# --------------------------------------------------------------------------
# unires/_project.py:99-190
# --------------------------------------------------------------------------
def proj_apply(operator, dat, po, method='super-resolution', bound='zero',
               interpolation='linear'):
    """A, At, AtA or 'none' on (1,1,X,Y,Z) data."""
    if operator not in ['A', 'At', 'AtA', 'none']:
        raise ValueError('Undefined operator')
    if method not in ['denoising', 'super-resolution']:
        raise ValueError('Undefined method')
    if operator == 'none':
        return dat
    mat, dim = proj_matrix(po, method)
    ratio, smo_ker, scl, dim_thick = po.ratio, po.smo_ker.to(dat.dtype), po.scl, po.dim_thick
    conv = lambda x: F.conv3d(x, smo_ker, stride=ratio)                 # :153
    conv_t = lambda x: F.conv_transpose3d(x, smo_ker, stride=ratio)     # :154
    grid = affine_grid(mat.type(dat.dtype), dim)[None]                  # :159
    kw = dict(bound=bound, extrapolate=False, interpolation=interpolation)
    if method == 'super-resolution':
        if operator == 'A':                                             # :163-167
            dat = conv(grid_pull(dat, grid, **kw))
            if scl != 0:
                dat = apply_scaling(dat, scl, dim_thick)
        elif operator == 'At':                                          # :168-172
            if scl != 0:
                dat = apply_scaling(dat, scl, dim_thick)
            dat = grid_push(conv_t(dat), grid, shape=po.dim_y, **kw)
        else:                                                           # :173-179
            dat = conv(grid_pull(dat, grid, **kw))
            if scl != 0:
                dat = apply_scaling(dat, 2 * scl, dim_thick)
            dat = grid_push(conv_t(dat), grid, shape=po.dim_y, **kw)
    else:                                                               # :180-188
        if operator == 'A':
            dat = grid_pull(dat, grid, **kw)
        elif operator == 'At':
            dat = grid_push(dat, grid, shape=po.dim_y, **kw)
        else:
            dat = grid_push(grid_pull(dat, grid, **kw), grid, shape=po.dim_y, **kw)
    return dat


# --------------------------------------------------------------------------
# unires/_project.py:300-317
# --------------------------------------------------------------------------
def DtD(dat, vx_y, bound='zero', diff='forward'):
    return im_divergence(im_gradient(dat, vx=vx_y, bound=bound, which=diff),
                         vx=vx_y, bound=bound, which=diff)


# --------------------------------------------------------------------------
# unires/_project.py:54-96
# --------------------------------------------------------------------------
def proj(operator, dat, x, y, method='super-resolution', do=True, rho=1, n=0, vx_y=None,
         interpolation='linear', bound='zero', diff='forward'):
    """x is the list of repeats of ONE channel; y the channel's output struct."""
    if operator == 'AtA':
        if not do:
            operator = 'none'
        dat = dat[None, None]
        dat_p = x[n].tau * proj_apply(operator, dat, x[n].po, method, bound, interpolation)
        for n1 in range(1, len(x)):
            dat_p = dat_p + x[n1].tau * proj_apply(operator, dat, x[n1].po, method, bound,
                                                   interpolation)
        dat_p = dat_p[0, 0]
        dat_p = dat_p + rho * y.lam ** 2 * DtD(dat[0, 0], vx_y=vx_y, bound=bound, diff=diff)
    else:
        if not do:
            operator = 'none'
        dat_p = proj_apply(operator, dat[None, None], x[n].po, method, bound, interpolation)[0, 0]
    return dat_p


# --------------------------------------------------------------------------
# unires/_project.py:27-51
# --------------------------------------------------------------------------
def check_adjoint(po, method, dtype=torch.float64):
    """<Ay,x> - <Atx,y> with seed-0 uniform inputs (returned, not printed)."""
    torch.manual_seed(0)
    x = torch.rand((1, 1) + tuple(po.dim_x), dtype=dtype)
    y = torch.rand((1, 1) + tuple(po.dim_y), dtype=dtype)
    Ay = proj_apply('A', y, po, method=method)
    Atx = proj_apply('At', x, po, method=method)
    return (torch.sum(Ay * x, dtype=torch.float64) - torch.sum(Atx * y, dtype=torch.float64)).item()


# --------------------------------------------------------------------------
# unires/_update.py:35-64
# --------------------------------------------------------------------------
def step_size(x, y, rho=None, rho_scl=1.0):
    if rho is not None:
        return torch.tensor(rho, dtype=torch.float32)
    all_lam = torch.tensor([float(yc.lam) for yc in y], dtype=torch.float32)
    all_tau = torch.tensor([float(xn.tau) for xc in x for xn in xc], dtype=torch.float32)
    return rho_scl * torch.sqrt(torch.mean(all_tau)) / torch.mean(all_lam)


# --------------------------------------------------------------------------
# unires/_update.py:118-152  (the y-update block of _update_admm)
# --------------------------------------------------------------------------
def y_rhs(xc, yc, zc, wc, rho, vx_y, method, do_proj):
    """b = sum_n tau_n At x_n - lam * Dt(w - rho z)   (:124-133)."""
    tmp = torch.zeros_like(yc.dat)
    for n in range(len(xc)):
        tmp += xc[n].tau * proj('At', xc[n].dat, xc, yc, method=method, do=do_proj, n=n)
    div = wc - rho * zc
    div = im_divergence(div, vx=vx_y)
    tmp -= yc.lam * div
    return tmp


def precond(xc, yc, rho, method):
    """Jacobi preconditioner x -> x / M  (unires/_update.py:80-102; commented out at :136):
    M = tau * AtA(1) + 2 rho lam^2 sum(1 / vx^2), one repeat per contrast only."""
    if len(xc) != 1:
        raise ValueError('CG pre-conditioning only supports one repeat per contrast.')   # :84-85
    lam = yc.lam
    vx = voxel_size(yc.mat).float()                                                    # :90
    M = xc[0].tau * proj_apply('AtA', torch.ones(tuple(yc.dim), dtype=torch.float32)[None, None],
                               xc[0].po, method=method)                                # :92-95
    M += 2 * rho * lam ** 2 * vx.square().reciprocal().sum()                            # :97
    M = M[0, 0]
    return lambda v: v / M                                                              # :100


def fft_precond(xc, yc, rho, method, do_proj=True):
    """Build-side extension (NOT in the reference): x -> IFFT(FFT(x) / (a + sum_d c_d lam_d)),
    the exact inverse of the circulant operator a I + rho lam^2 sum_d L_d / vx_d^2 with periodic
    second differences; a = mean diagonal of sum_n tau_n AtA_n (tau for A = I)."""
    import math
    dim = tuple(yc.dim)
    vx = voxel_size(yc.mat).float()
    if do_proj:
        acc = torch.zeros(dim)
        for xn in xc:
            acc += xn.tau * proj_apply('AtA', torch.ones(dim)[None, None], xn.po, method=method)[0, 0]
        a = float(acc.double().mean())
    else:
        a = float(sum(float(xn.tau) for xn in xc))
    c = [float(rho) * float(yc.lam) ** 2 / float(vx[d]) ** 2 for d in range(3)]
    lam = [2.0 - 2.0 * torch.cos(2.0 * math.pi * torch.arange(n, dtype=torch.float64) / n) for n in dim]
    den = a + c[0] * lam[0][:, None, None] + c[1] * lam[1][None, :, None] \
        + c[2] * lam[2][None, None, :dim[2] // 2 + 1]
    return lambda v: torch.fft.irfftn(torch.fft.rfftn(v.double()) / den, s=dim).float()


def update_y(x, y, z, w, rho, method, do_proj, cgs_max_iter=20, cgs_tol=1e-3,
             return_info=False, jacobi=False, fft=False):
    """In-place CG update of every y[c].dat; identity preconditioner (:136-137),
    stop='max_gain' (:145).  ``jacobi`` enables the line the reference keeps commented
    out (:136)."""
    vx_y = voxel_size(y[0].mat).float()                                   # :111
    info = []
    for c in range(len(x)):
        tmp = y_rhs(x[c], y[c], z[c], w[c], rho, vx_y, method, do_proj)
        lhs = lambda dat, c=c: proj('AtA', dat, x[c], y[c], method=method, do=do_proj,
                                    rho=rho, vx_y=vx_y)                   # :140-141
        pre = precond(x[c], y[c], rho, method) if jacobi else (lambda r: r)   # :136-137
        if fft:
            pre = fft_precond(x[c], y[c], rho, method, do_proj)
        _, n_it, obj = cg(A=lhs, b=tmp, x=y[c].dat, max_iter=cgs_max_iter, stop='max_gain',
                          inplace=True, precond=pre, tolerance=cgs_tol,
                          return_info=True)                               # :142-148
        info.append((n_it, obj))
    return (y, info) if return_info else y


# --------------------------------------------------------------------------
# unires/_update.py:270-393, 430-445  (even/odd slice-scaling Gauss-Newton)
# --------------------------------------------------------------------------
def even_odd(dat, which, dim):
    """'odd' = [::2], 'even' = [1::2] along ``dim`` (:430-445)."""
    sl = [slice(None)] * 3
    sl[dim] = slice(0, None, 2) if which == 'odd' else slice(1, None, 2)
    return dat[tuple(sl)]


def update_scaling(x, y, method='super-resolution', max_niter_gn=1, num_linesearch=4):
    """_update_scaling, restated line by line - including the line search that rescales the
    already rescaled ``dat_y`` after a rejected step (:362-366).  ``rigid`` is read from
    ``po.rigid`` (the reference recomputes it from rigid_q, :301; same matrix).  Returns
    (x, sll); ``x[c][n].po.scl`` is updated (float64 0-d tensor after the first step)."""
    from .nitorch_restated import affine_grid, grid_pull
    import torch.nn.functional as F
    sll = torch.tensor(0, dtype=torch.float64)
    ll = torch.tensor(0, dtype=torch.float64)
    for c in range(len(x)):
        for n_x in range(len(x[c])):
            xn = x[c][n_x]
            if getattr(xn, 'ct', False):
                continue                                                          # :288-290
            po = xn.po
            dim_thick, tau = po.dim_thick, xn.tau
            scl = torch.as_tensor(po.scl)
            mat = torch.linalg.solve(po.mat_y, po.rigid.mm(po.mat_yx))            # :302
            dat_x = xn.dat
            msk = dat_x != 0
            xo = even_odd(dat_x, 'odd', dim_thick)[even_odd(msk, 'odd', dim_thick)]
            xe = even_odd(dat_x, 'even', dim_thick)[even_odd(msk, 'even', dim_thick)]
            mo, me = even_odd(msk, 'odd', dim_thick), even_odd(msk, 'even', dim_thick)
            grid = affine_grid(mat.type(torch.float32), po.dim_yx)
            dat_y = grid_pull(y[c].dat[None, None], grid[None], bound='zero',
                              interpolation='linear', extrapolate=False)
            dat_y = F.conv3d(dat_y, po.smo_ker, stride=po.ratio)[0, 0]           # :320
            dat_y = apply_scaling(dat_y[None, None], scl, dim_thick)[0, 0]       # :322
            for _ in range(max_niter_gn):
                ll = 0.5 * tau * torch.sum((dat_x[msk] - dat_y[msk]) ** 2, dtype=torch.float64)
                yo = even_odd(dat_y, 'odd', dim_thick)[mo]
                ye = even_odd(dat_y, 'even', dim_thick)[me]
                gr = tau * (torch.sum(ye * (xe - ye), dtype=torch.float64)
                            - torch.sum(yo * (xo - yo), dtype=torch.float64))   # :340-341
                Hes = tau * (torch.sum(ye ** 2, dtype=torch.float64)
                             + torch.sum(yo ** 2, dtype=torch.float64))          # :344-345
                Update = gr / Hes
                old_scl, old_ll = scl.clone(), ll.clone()
                armijo = torch.tensor(1.0, dtype=old_scl.dtype)
                if num_linesearch == 0:
                    scl = old_scl - armijo * Update
                else:
                    for _ls in range(num_linesearch):
                        scl = old_scl - armijo * Update
                        dat_y = apply_scaling(dat_y[None, None], scl - old_scl, dim_thick)[0, 0]
                        ll = 0.5 * tau * torch.sum((dat_x[msk] - dat_y[msk]) ** 2,
                                                   dtype=torch.float64)
                        if ll < old_ll:
                            break
                        scl, ll = old_scl, old_ll
                        armijo = armijo * 0.5
            po.scl = scl                                                          # :389
            sll = sll + ll
    return x, sll


# --------------------------------------------------------------------------
# unires/_update.py:160-193  (z- and w-updates)
# --------------------------------------------------------------------------
def update_zw(y, z, w, rho, alpha=1.0):
    vx_y = voxel_size(y[0].mat).float()
    tiny = torch.tensor(1e-7, dtype=torch.float32)
    one = torch.tensor(1.0, dtype=torch.float32)
    rho = torch.as_tensor(rho, dtype=torch.float32)
    z_old = z.clone() if alpha != 1 else None
    C = len(y)

    def _Dy(c):
        Dy = y[c].lam * im_gradient(y[c].dat, vx=vx_y)
        if alpha != 1:
            Dy = alpha * Dy + (one - alpha) * z_old[c]
        return Dy

    tmp = torch.zeros_like(y[0].dat)
    for c in range(C):
        tmp += torch.sum((w[c] / rho + _Dy(c)) ** 2, dim=0)
    tmp.sqrt_()
    tmp = ((tmp - one / rho).clamp_min(0)) / (tmp + tiny)
    for c in range(C):
        Dy = _Dy(c)
        for d in range(3):
            z[c, d] = tmp * (w[c, d] / rho + Dy[d])
    for c in range(C):
        w[c] += rho * (_Dy(c) - z[c])
    return z, w, tmp


# --------------------------------------------------------------------------
# unires/_update.py:396-427  (objective)
# --------------------------------------------------------------------------
def compute_nll(x, y, method, do_proj, sum_dtype=torch.float64):
    vx_y = voxel_size(y[0].mat).float()
    nll_xy = torch.tensor(0, dtype=torch.float64)
    nll_y = None
    for c in range(len(x)):
        for n in range(len(x[c])):
            msk = x[c][n].dat != 0
            Ay = proj('A', y[c].dat, x[c], y[c], n=n, method=method, do=do_proj)
            nll_xy = nll_xy + 0.5 * x[c][n].tau * torch.sum(
                (x[c][n].dat[msk] - Ay[msk]) ** 2, dtype=sum_dtype)
        Dy = y[c].lam * im_gradient(y[c].dat, vx=vx_y)
        nll_y = torch.sum(Dy ** 2, dim=0) if nll_y is None else nll_y + torch.sum(Dy ** 2, dim=0)
    nll_y = torch.sum(torch.sqrt(nll_y), dtype=sum_dtype)
    return nll_xy + nll_y, nll_xy, nll_y


# --------------------------------------------------------------------------
# unires/_core.py:371-399  (initial guess: trilinear reslice)
# --------------------------------------------------------------------------
def init_y_dat(x, y):
    """y[c].dat = average over repeats of the inputs pulled into the mean space,
    clamped to each input's range; zeros where no input covers the voxel."""
    dim_y = y[0].dim
    mat_y = y[0].mat
    for c in range(len(x)):
        dat_y = torch.zeros(dim_y, dtype=torch.float32)
        sm = torch.zeros_like(dat_y)
        for n in range(len(x[c])):
            dat = x[c][n].dat[None, None]
            mat = torch.linalg.solve(x[c][n].mat, mat_y)            # :385  mat_x \ mat_y
            grid = affine_grid(mat.type(dat.dtype), dim_y)          # :386
            mn, mx = torch.min(dat), torch.max(dat)
            dat = grid_pull(dat, grid[None], bound='zero', extrapolate=False, interpolation=1)
            dat[dat < mn] = mn
            dat[dat > mx] = mx
            sm = sm + (dat[0, 0] > 0)
            dat_y = dat_y + dat[0, 0]
        sm[sm == 0] = 1.0
        y[c].dat = dat_y / sm
    return y


# --------------------------------------------------------------------------
# unires/run.py:24-207 (fit) and unires/_core.py:288-307 (_get_sched)
# --------------------------------------------------------------------------
def get_sched(N, reg_scl=4.0, sched_num=3):
    """Coarse-to-fine schedule of the regularisation scaling: 4 -> [32, 16, 8, 4]; [4] if N == 1."""
    if sched_num < 0 or N == 1:
        sched_num = 0
    scl = torch.as_tensor(reg_scl, dtype=torch.float32).reshape(1)
    sched = (torch.tensor(2.0) ** torch.arange(0, 32, step=1, dtype=torch.float32)).flip(dims=(0,))
    ix = torch.min((sched - scl).abs(), dim=0)[1]
    return torch.cat((sched[ix - sched_num:ix], scl))


def fit(x, y, method, do_proj, max_iter=512, tolerance=1e-4, reg_scl=4.0, sched_num=3,
        scaling=False, cgs_max_iter=20, cgs_tol=1e-3, alpha=1.0, unified_rigid=False,
        rigid_mod=1, rigid_basis=None):
    """The iteration loop of fit() (run.py:56-143) on in-memory structs: lambda scaling,
    rho, ADMM iterations, convergence countdowns, optional scaling update, coarse-to-fine
    switch.  y[c].lam0 must hold the unscaled regularisation.  Returns (y, obj, n_iter, sched)."""
    from .nitorch_restated import get_gain
    N = sum(len(xn) for xn in x)
    sched = get_sched(N, reg_scl, sched_num)
    cnt_scl = 0
    for c in range(len(x)):
        y[c].lam = sched[cnt_scl] * y[c].lam0                                     # :60-61
    rho = step_size(x, y)                                                         # :65
    z = torch.zeros((len(y), 3) + tuple(y[0].dat.shape), dtype=torch.float32)
    w = torch.zeros_like(z)
    obj = torch.zeros(max_iter, 3, dtype=torch.float64)
    cnt_scl_iter = 0
    countdown0 = countdown1 = 6
    n_done = 0
    for n_iter in range(max_iter):
        y = update_y(x, y, z, w, rho, method, do_proj, cgs_max_iter=cgs_max_iter, cgs_tol=cgs_tol)
        if tolerance > 0:
            obj[n_iter, 0], obj[n_iter, 1], obj[n_iter, 2] = compute_nll(x, y, method, do_proj)
        z, w, _ = update_zw(y, z, w, rho, alpha=alpha)
        n_done = n_iter + 1
        gain = get_gain(obj[:n_iter + 1, 0], monotonicity='decreasing')            # :100
        if cnt_scl >= (sched.numel() - 1) and cnt_scl_iter > 20 \
                and ((gain.abs() < tolerance) or (n_iter >= (max_iter - 1))):      # :103-104
            countdown0 -= 1
            if countdown0 == 0:
                break
        else:
            countdown0 = 6
        if scaling:
            x, _ = update_scaling(x, y, method=method, max_niter_gn=1, num_linesearch=6)  # :119
        if unified_rigid and n_iter > 0 and (n_iter % rigid_mod) == 0:              # :127-132
            x, _ = update_rigid(x, y, method, rigid_basis, mean_correct=False, max_niter_gn=1,
                                num_linesearch=6)
        if cnt_scl + 1 < len(sched) and cnt_scl_iter > 16 and gain.abs() < 1e-3:   # :139
            countdown1 -= 1
            if countdown1 == 0:
                cnt_scl_iter = 0
                cnt_scl += 1
                for c in range(len(x)):
                    y[c].lam = sched[cnt_scl] * y[c].lam0
                rho = step_size(x, y)
        else:
            countdown1 = 6
        cnt_scl_iter += 1
    return y, obj[:n_done], n_done, sched


# --------------------------------------------------------------------------
# unires/_update.py:198-266, 448-538, 541-710  (unified rigid registration)
# --------------------------------------------------------------------------
def affine_basis_se3():
    """A basis of se(3), (6,4,4) float64  [nitorch affine_basis('SE') recalled: translations,
    then rotations].  Deliberately NOT the product's ordering / signs: rigid matrices and
    log-likelihoods of a Gauss-Newton step do not depend on the basis (only q does), and the
    parity tests check exactly that."""
    B = torch.zeros(6, 4, 4, dtype=torch.float64)
    for d in range(3):
        B[d, d, 3] = 1.0
    B[3, 0, 1], B[3, 1, 0] = 1.0, -1.0
    B[4, 0, 2], B[4, 2, 0] = 1.0, -1.0
    B[5, 1, 2], B[5, 2, 1] = 1.0, -1.0
    return B


def expm(q, basis, grad_X=False):
    """expm(sum q_i B_i) and, optionally, its derivative w.r.t. every q_i: (num_q, 4, 4)."""
    f = lambda v: torch.linalg.matrix_exp(torch.einsum('i,ijk->jk', v, basis))
    R = f(q)
    if not grad_X:
        return R
    with torch.enable_grad():
        J = torch.autograd.functional.jacobian(f, q)
    return R, J.permute(2, 0, 1)


def rigid_match(dat_x, dat_y, po, tau, rigid, method, CtC=None, diff=False):
    """_rigid_match (:448-538): ll, gr (dim,3), Hes (dim,6)."""
    import torch.nn.functional as F
    from .nitorch_restated import affine_grid, grid_grad, grid_pull
    if method == 'super-resolution':
        dim, mat = tuple(po.dim_yx), po.mat_yx
    else:
        dim, mat = tuple(po.dim_x), po.mat_x
    mat = torch.linalg.solve(po.mat_y, rigid.mm(mat))                              # :498
    grid = affine_grid(mat.type(torch.float32), dim)
    dat_yx = grid_pull(dat_y, grid[None], bound='zero', extrapolate=False)[0, 0]   # :502
    if method == 'super-resolution':
        dat_yx = F.conv3d(dat_yx[None, None], po.smo_ker, stride=po.ratio)[0, 0]
        if po.scl != 0:
            dat_yx = apply_scaling(dat_yx[None, None], po.scl, po.dim_thick)[0, 0]
    gr = Hes = None
    if diff:
        gr = grid_grad(dat_y, grid[None], bound='zero', extrapolate=False)[0, 0]   # :508
    msk = dat_x != 0
    ll = 0.5 * tau * torch.sum((dat_x[msk] - dat_yx[msk]) ** 2, dtype=torch.float64)
    if diff:
        d = dat_yx - dat_x
        msk = msk & (dat_yx != 0)
        d[~msk] = 0
        Hes = torch.zeros(dim + (6,), dtype=torch.float32)
        Hes[..., 0] = gr[..., 0] * gr[..., 0]
        Hes[..., 1] = gr[..., 1] * gr[..., 1]
        Hes[..., 2] = gr[..., 2] * gr[..., 2]
        Hes[..., 3] = gr[..., 0] * gr[..., 1]
        Hes[..., 4] = gr[..., 0] * gr[..., 2]
        Hes[..., 5] = gr[..., 1] * gr[..., 2]
        if method == 'super-resolution':
            Hes = Hes * CtC[..., None]
            d = F.conv_transpose3d(d[None, None], po.smo_ker, stride=po.ratio)[0, 0]   # :524
        gr = gr * d[..., None]
    return ll, gr, Hes


def update_rigid_channel(xc, yc, method, basis, max_niter_gn=1, num_linesearch=4, samp=0,
                         prof_ip=0, prof_tp=0, gap=0.0):
    """_update_rigid_channel (:541-710).  ``samp = 0``: xc[n].po is used where the reference
    rebuilds an identical one (:575-578).  ``samp > 0``: po is rebuilt with the sub-sampling branch
    of _proj_info and the low-res data resampled onto the decimated lattice (:589-593: grid_pull
    at the integer coordinates D_x u with interpolation 0 = a strided slice); the high-res side is
    never resampled (D_y stays None, see proj_info)."""
    import torch.nn.functional as F
    from .nitorch_restated import affine_grid
    num_q = basis.shape[0]
    lkp = [[0, 3, 4], [3, 1, 5], [4, 5, 2]]
    one = torch.tensor(1.0, dtype=torch.float64)
    sll = torch.tensor(0, dtype=torch.float64)
    for n_x in range(len(xc)):
        dat_x = xc[n_x].dat
        q = xc[n_x].rigid_q.clone()
        tau = xc[n_x].tau
        armijo = torch.tensor(1, dtype=q.dtype)
        po = xc[n_x].po
        if samp > 0:                                                                 # :575-593
            po0 = xc[n_x].po
            po = proj_info(po0.dim_y, po0.mat_y, po0.dim_x, po0.mat_x, rigid=po0.rigid, prof_ip=prof_ip,
                           prof_tp=prof_tp, gap=gap, scl=po0.scl, samp=samp)
            sk = [int(v) for v in torch.diagonal(po.D_x)[:3].tolist()]
            dat_x = dat_x[::sk[0], ::sk[1], ::sk[2]][:po.dim_x[0], :po.dim_x[1], :po.dim_x[2]]
        if method == 'super-resolution':
            dim, mat = tuple(po.dim_yx), po.mat_yx
        else:
            dim, mat = tuple(po.dim_x), po.mat_x
        dat_y = yc.dat[None, None]
        CtC = None
        if method == 'super-resolution':
            CtC = F.conv3d(torch.ones((1, 1) + dim), po.smo_ker, stride=po.ratio)
            CtC = F.conv_transpose3d(CtC, po.smo_ker, stride=po.ratio)[0, 0]         # :603-607
        id_x = affine_grid(torch.eye(4), dim)                                       # :610
        ll = torch.tensor(0, dtype=torch.float64)
        rigid = expm(q, basis)
        for _gn in range(max_niter_gn):
            rigid, d_rigid = expm(q, basis, grad_X=True)
            d_rigid_q = torch.zeros(4, 4, num_q, dtype=torch.float64)
            for i in range(num_q):
                d_rigid_q[:, :, i] = torch.linalg.solve(po.mat_y, d_rigid[i].mm(mat))   # :622
            gr = torch.zeros(num_q, 1, dtype=torch.float64)
            Hes = torch.zeros(num_q, num_q, dtype=torch.float64)
            ll, gr_m, Hes_m = rigid_match(dat_x, dat_y, po, tau, rigid, method, CtC=CtC, diff=True)
            dAff = []
            for i in range(num_q):
                dAff.append([])
                for d in range(3):
                    dAff[i].append(d_rigid_q[d, 0, i] * id_x[..., 0] + d_rigid_q[d, 1, i] * id_x[..., 1]
                                   + d_rigid_q[d, 2, i] * id_x[..., 2] + d_rigid_q[d, 3, i])
            for d in range(3):
                for i in range(num_q):
                    gr[i] += torch.sum(gr_m[..., d] * dAff[i][d], dtype=torch.float64)
            for d1 in range(3):
                for d2 in range(3):
                    for i1 in range(num_q):
                        tmp1 = Hes_m[..., lkp[d1][d2]] * dAff[i1][d1]
                        for i2 in range(i1, num_q):
                            Hes[i1, i2] += torch.sum(tmp1 * dAff[i2][d2], dtype=torch.float64)
            for i1 in range(num_q):
                for i2 in range(i1 + 1, num_q):
                    Hes[i2, i1] = Hes[i1, i2]
            Update = torch.linalg.solve(Hes, gr)[:, 0]                               # :660
            old_ll, old_q, old_rigid = ll.clone(), q.clone(), rigid.clone()
            if num_linesearch == 0:
                q = old_q - armijo * Update
                rigid = expm(q, basis)
            else:
                for _ls in range(num_linesearch):
                    q = old_q - armijo * Update
                    rigid = expm(q, basis)
                    ll = rigid_match(dat_x, dat_y, po, tau, rigid, method)[0]
                    if ll < old_ll:
                        armijo = torch.min(1.25 * armijo, one)
                        break
                    ll, q, rigid = old_ll, old_q, old_rigid
                    armijo = armijo * 0.5
        xc[n_x].rigid_q = q
        xc[n_x].po.rigid = rigid
        sll = sll + ll
    return xc, sll


def update_rigid(x, y, method, basis, mean_correct=True, max_niter_gn=1, num_linesearch=4, samp=0):
    """_update_rigid (:198-266)."""
    sll = torch.tensor(0, dtype=torch.float64)
    for c in range(len(x)):
        x[c], sllc = update_rigid_channel(x[c], y[c], method, basis, max_niter_gn=max_niter_gn,
                                          num_linesearch=num_linesearch, samp=samp)
        sll = sll + sllc
    if mean_correct:
        qs = [xn.rigid_q for xc in x for xn in xc]
        mean_q = torch.stack(qs).sum(0) / float(len(qs))
        for xc in x:
            for xn in xc:
                xn.rigid_q = xn.rigid_q - mean_q
                xn.po.rigid = expm(xn.rigid_q, basis)
    return x, sll
