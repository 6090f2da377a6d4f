"""ADMM updates - host-side mirror of the y-update block of unires/_update.py
(:105-152), plus _step_size (:35-64) and _admm_aux (:17-32)."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, f3, i3
from ._ops import _ptr, _stream, on_device
from ._host import Pacer
from ._plan import cg_many
from ._project import _channel_plan, _proj
from .spatial import voxel_size


_PACERS = {}


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


class _Precond:
    """Callable x -> x / M like the lambda unires/_update.py:100 returns; carries the plan
    that holds the device copy of M so cg() can run the preconditioned iteration on device."""

    def __init__(self, plan, M, mode):
        self.plan, self.M, self.mode = plan, M, mode

    def __call__(self, v):
        return v if self.M is None else v / self.M


@on_device
def _precond(x, y, rho, sett):
    """Compute CG preconditioner (unires/_update.py:80-102): Jacobi,
    M = tau AtA(1) + 2 rho lam^2 sum(1/vx^2) for one channel (x = x[c], y = y[c])."""
    if len(x) != 1:
        raise ValueError('CG pre-conditioning only supports one repeat per contrast.')
    plan = _channel_plan(x, y, sett.method, sett.do_proj, voxel_size(y.mat).float())
    M = torch.empty(tuple(y.dim), dtype=torch.float32, device=y.dat.device)
    plan.precond_build(float(rho), float(y.lam), mode='jacobi', out=M)
    return _Precond(plan, M, 'jacobi')


@on_device
def _update_y(x, y, z, w, rho, tmp, sett, info=None):
    """UPDATE: y  (unires/_update.py:118-152).  Per channel: assemble
    b = sum_n tau_n At x_n - lam Dt(w - rho z) and solve
    (sum tau AtA + rho lam^2 DtD) y = b by CG, in place on y[c].dat.

    The channels do not couple inside the y-update (no cross-channel term in
    :122-150), so each channel's RHS + CG is enqueued on its own HIP stream: the
    issue-bound push kernel of one channel overlaps the HBM-bound vector kernels of
    another.  Results are identical to the sequential loop; ``tmp`` (the reference's
    shared RHS buffer) holds channel 0's b, the other channels use per-plan buffers.

    With ``sett.cgs_tol > 0`` (the reference's default) the solves can stop early: the library feeds them to
    the device chunk by chunk and this call returns when the LAST chunk is enqueued - the calling thread stays
    in the feeder (napping between looks at a host-mapped progress word) for about as long as the solves run;
    host / device overlap beyond that needs ``cgs_tol = 0``.  Under stream capture the whole solve is enqueued
    instead (kernels after convergence return at entry).
    """
    vx_y = voxel_size(y[0].mat).float()
    rho = float(rho)
    sync = info is not None
    C = len(x)
    concurrent = C > 1 and not sync and channel_streams_on(sett, y[0].dat)
    pre = getattr(sett, 'cgs_precond', 'none')
    if not concurrent:
        for c in range(C):
            plan = _channel_plan(x[c], y[c], sett.method, sett.do_proj, vx_y)
            plan.set_concurrency(1)
            lam = float(y[c].lam)
            if getattr(sett, 'cache_atx', True) and not sync:
                plan.rhs_cached([xn.dat for xn in x[c]], w[c], z[c], rho, lam, out=tmp)
            else:
                plan.rhs([xn.dat for xn in x[c]], w[c], z[c], rho, lam, out=tmp)
            if pre in ('jacobi', 'fft'):
                plan.precond_build(rho, lam, mode=pre)
            res = plan.cg(tmp, y[c].dat, rho, lam, max_iter=sett.cgs_max_iter,
                          tolerance=sett.cgs_tol, stop=sett.cgs_stop, sync=sync, precond=pre)
            if sync:
                info.append(res)
        return y
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)
    streams = _side_streams(y[0].dat.device, C)
    plans, bs = [], []
    for c in range(C):
        plan = _channel_plan(x[c], y[c], sett.method, sett.do_proj, vx_y)
        plan.set_concurrency(C)  # (its persistent kernels leave the other channels' kernels room on the CUs)
        lam = float(y[c].lam)
        b = tmp if c == 0 else plan.rhs_buffer(tmp)
        with torch.cuda.stream(streams[c]):
            streams[c].wait_event(ready)
            if getattr(sett, 'cache_atx', True):
                plan.rhs_cached([xn.dat for xn in x[c]], w[c], z[c], rho, lam, out=b)
            else:
                plan.rhs([xn.dat for xn in x[c]], w[c], z[c], rho, lam, out=b)
            if pre in ('jacobi', 'fft'):
                plan.precond_build(rho, lam, mode=pre)
        plans.append(plan)
        bs.append(b)
    # the solves of all channels from one call: with a tolerance the library feeds them chunk by chunk
    # from one host loop (unires_cg_solve_many), each on its channel's stream
    cg_many(plans, bs, [yc.dat for yc in y], rho, [float(yc.lam) for yc in y], streams,
            max_iter=sett.cgs_max_iter, tolerance=sett.cgs_tol, stop=sett.cgs_stop, precond=pre)
    for c in range(C):
        main.wait_stream(streams[c])
    return y


# struct.settings.channel_streams = 'auto': the channels of a y-update go to separate HIP streams whenever there is
# more than one.  (Rounds 3 - 5 stopped at 10 M voxels: at 256^3 three streams bought nothing - every matvec kernel
# fills the chip - and cost 6 % before the runtime's signal pool was raised.  Since round 6 a plan that is told about
# its neighbours, `ChannelPlan.set_concurrency`, sizes its persistent kernels for them: profiles/r06_overlap_scan.txt,
# +6 % CG it/s at 256^3 x 3, +5 % at 384^3 x 4, +24 % at 181 x 217 x 181 x 3.)  None: no upper bound.
CHANNEL_STREAMS_MAX_VOXELS = None


def channel_streams_on(sett, dat):
    cs = getattr(sett, 'channel_streams', 'auto')
    if cs == 'auto':
        return CHANNEL_STREAMS_MAX_VOXELS is None or dat.numel() < CHANNEL_STREAMS_MAX_VOXELS
    return bool(cs)


_STREAMS = {}


def _side_streams(device, n):
    key = (device.type, device.index)
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _chan_args(y):
    for yc in y:
        if not yc.dat.is_cuda or yc.dat.dtype != torch.float32 or not yc.dat.is_contiguous():
            raise RuntimeError('unires_amd: y[c].dat must be contiguous float32 CUDA/HIP tensors')
    ptrs = (C.c_void_p * len(y))(*[yc.dat.data_ptr() for yc in y])
    lams = (C.c_float * len(y))(*[float(yc.lam) for yc in y])
    return ptrs, lams


@on_device
def _update_zw(y, z, w, rho, tmp, sett):
    """UPDATE z and w  (unires/_update.py:160-193): joint-TV shrinkage and dual ascent,
    two fused kernels per channel instead of three im_gradient passes and a dozen
    temporaries.  ``tmp`` receives the shrinkage image, as in the reference."""
    for t in (z, w):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() \
                or tuple(t.shape) != (len(y), 3) + tuple(y[0].dim):
            raise ValueError('unires_amd: z/w must be contiguous (C,3,X,Y,Z) float32 device tensors')
    if not tmp.is_contiguous() or tuple(tmp.shape) != tuple(y[0].dim):
        raise ValueError('unires_amd: tmp must be a contiguous (X,Y,Z) tensor')
    vx = [float(v) for v in voxel_size(y[0].mat).tolist()]
    ptrs, lams = _chan_args(y)
    check(_lib.load().unires_zw_update(ptrs, lams, len(y), i3(y[0].dim), f3(vx), float(rho),
                                       float(sett.alpha), _ptr(z), _ptr(w), _ptr(tmp), _stream()))
    return z, w, tmp


@on_device
def _compute_nll(x, y, sett, rho, sum_dtype=torch.float64):
    """Negative model log-likelihood (unires/_update.py:396-427); returns 0-d float64
    device tensors (nll_yx, nll_xy, nll_y) without synchronising."""
    dev = y[0].dat.device
    lib = _lib.load()
    vx = [float(v) for v in voxel_size(y[0].mat).tolist()]
    nll_xy = torch.zeros((), dtype=torch.float64, device=dev)
    sse = torch.zeros((), dtype=torch.float64, device=dev)
    for c in range(len(x)):
        for n in range(len(x[c])):
            Ay = _proj('A', y[c].dat, x[c], y[c], n=n, method=sett.method, do=sett.do_proj)
            xd = x[c][n].dat.contiguous()
            check(lib.unires_masked_sse(_ptr(xd), _ptr(Ay), xd.numel(), _ptr(sse), _stream()))
            nll_xy = nll_xy + 0.5 * float(x[c][n].tau) * sse
    ptrs, lams = _chan_args(y)
    nll_y = torch.zeros((), dtype=torch.float64, device=dev)
    check(lib.unires_nll_prior(ptrs, lams, len(y), i3(y[0].dim), f3(vx), _ptr(nll_y), _stream()))
    return nll_xy + nll_y, nll_xy, nll_y


@on_device
def _update_scaling(x, y, sett, max_niter_gn=1, num_linesearch=4, verbose=0):
    """Updates the even/odd slice scaling parameter of every observation by Gauss-Newton
    (unires/_update.py:270-393).  ``A y`` comes from the fused pull+conv kernel, the five masked
    float64 sums of a step (ll, gradient and Hessian terms) from ONE reduction kernel instead
    of ten masked-index passes; the step logic - including the reference's line search, which
    rescales the already rescaled ``dat_y`` after a rejected step (:362-366) - runs on the host
    with one 40-byte read-back per evaluation.  ``po.rigid`` is used where the reference
    recomputes the same matrix from ``rigid_q`` (:301).  Returns (x, sll)."""
    from ._project import _apply_scaling
    if sett.method != 'super-resolution':
        raise ValueError('_update_scaling needs the super-resolution projection '
                         '(slice profile + slice scaling)')
    lib = _lib.load()
    dev = y[0].dat.device
    sll = torch.zeros((), dtype=torch.float64, device=dev)
    sums = torch.empty(5, dtype=torch.float64, device=dev)
    for c in range(len(x)):
        for n_x in range(len(x[c])):
            xn = x[c][n_x]
            if xn.ct:  # do not optimise scaling for CT data (:288-290)
                continue
            po = xn.po
            dim_thick, tau, scl = int(po.dim_thick), float(xn.tau), float(po.scl)
            plan = _channel_plan(x[c], y[c], sett.method, sett.do_proj)
            dat_x = xn.dat.contiguous()
            dat_y = plan.proj_apply(n_x, 'A', y[c].dat)  # pull + conv + S(scl)  (:316-322)

            def evaluate(ay):
                check(lib.unires_scaling_sums(_ptr(dat_x), _ptr(ay), i3(dat_x.shape), dim_thick,
                                              _ptr(sums), _stream()))
                return sums.tolist()  # one small device-to-host copy

            ll = 0.0
            for _ in range(max_niter_gn):
                s = evaluate(dat_y)
                ll = 0.5 * tau * s[0]
                gr = tau * (s[1] - s[2])
                hes = tau * (s[3] + s[4])
                update = float(torch.tensor(gr, dtype=torch.float64) / torch.tensor(hes, dtype=torch.float64))  # IEEE: 0/0 -> nan, like the reference's tensors
                old_scl, old_ll, armijo = scl, ll, 1.0
                if num_linesearch == 0:
                    scl = old_scl - armijo * update
                else:
                    for _ls in range(num_linesearch):
                        scl = old_scl - armijo * update
                        dat_y = _apply_scaling(dat_y, scl - old_scl, dim_thick)
                        ll = 0.5 * tau * evaluate(dat_y)[0]
                        if ll < old_ll:
                            break
                        scl, ll = old_scl, old_ll
                        armijo *= 0.5
                if verbose >= 1:
                    print('c={}, n={} | exp(s)={:.5f} ll={:.2f}'.format(c, n_x, torch.tensor(scl).exp(), ll))
            po.scl = scl  # the next _channel_plan() call rebuilds the operator with it
            sll = sll + ll
    return x, sll


@on_device
def _update_admm(x, y, z, w, rho, tmp, obj, n_iter, sett, info=None):
    """One ADMM iteration (unires/_update.py:105-195): y-update (CG), objective,
    z-update, w-update - same order, same in-place semantics, `tmp` returned as the
    joint-TV shrinkage image."""
    y = _update_y(x, y, z, w, rho, tmp, sett, info)
    if obj is not None and sett.tolerance > 0:
        obj[n_iter, 0], obj[n_iter, 1], obj[n_iter, 2] = _compute_nll(x, y, sett, rho)
    z, w, tmp = _update_zw(y, z, w, rho, tmp, sett)
    # An iteration is enqueue-only (no read-back): a host loop would run ahead until the hardware queue is
    # full and then spin inside every launch call.  The pacer parks the thread - sleeping between looks at a
    # stream mark (_host.StreamMark) - until the device is at most `host_pace` iterations behind (settings.host_pace; 0: off).
    depth = int(getattr(sett, 'host_pace', 2) or 0)
    if depth > 0 and y[0].dat.is_cuda:
        key = (y[0].dat.device.index, depth)
        pacer = _PACERS.get(key)
        if pacer is None:
            pacer = _PACERS[key] = Pacer(depth)
        pacer.step()
    return y, z, w, tmp, obj
