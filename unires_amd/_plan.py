"""ChannelPlan - Python handle of ``unires_plan_t``: the per-channel operator
``sum_n tau_n A_n^T A_n + rho lam^2 D^T D`` with all its device workspace."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (REGIME_DENOISE, REGIME_IDENTITY, REGIME_SUPERRES, Repeat, check, f3, i3)
from ._ops import FOV_TOL, _ptr, _stream, _vol, on_device
from .spatial import _m12, voxel_size


def proj_matrix(po, method):
    """float64 mat_y \\ (rigid @ mat_yx | mat_x) and the grid dims
    (unires/_project.py:145-150)."""
    if method == 'super-resolution':
        return torch.linalg.solve(po.mat_y, po.rigid.mm(po.mat_yx)), tuple(po.dim_yx)
    if method == 'denoising':
        return torch.linalg.solve(po.mat_y, po.rigid.mm(po.mat_x)), tuple(po.dim_x)
    raise ValueError('Undefined method')


def regime_of(method, do_proj):
    if method not in ('denoising', 'super-resolution'):
        raise ValueError('Undefined method')
    if not do_proj:
        return REGIME_IDENTITY
    return REGIME_SUPERRES if method == 'super-resolution' else REGIME_DENOISE


def _repeat_desc(po, tau, method, regime, keep):
    r = Repeat()
    r.tau = float(tau)
    if regime == REGIME_IDENTITY:
        return r
    mat, dim_g = proj_matrix(po, method)
    r.dim_x = i3(po.dim_x)
    r.dim_g = i3(dim_g)
    r.M = _lib.c_f32x12(*_m12(mat).tolist())
    if regime == REGIME_SUPERRES:
        taps = [np.ascontiguousarray(np.asarray(k, dtype=np.float32)) for k in po.smo_ker_1d]
        keep.extend(taps)
        r.ratio = i3(po.ratio)
        r.ntaps = i3([len(k) for k in taps])
        r.taps = _lib.c_fptrx3(*[k.ctypes.data_as(C.POINTER(C.c_float)) for k in taps])
        r.scl = float(po.scl)
        r.dim_thick = int(po.dim_thick)
    else:
        r.ratio = i3((1, 1, 1))
        r.ntaps = i3((1, 1, 1))
    return r


class ChannelPlan:
    """One channel's fused operator. ``xs``: list of (po, tau) per repeat."""

    def __init__(self, dim_y, vx_y, repeats, method, do_proj, fov_tol=FOV_TOL, device=None):
        self.lib = _lib.load()
        # the plan's workspace lives on ONE device: the one current when it is created
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        with torch.cuda.device(self.device):
            self._create(dim_y, vx_y, repeats, method, do_proj, fov_tol)

    def _create(self, dim_y, vx_y, repeats, method, do_proj, fov_tol):
        self.dim_y = tuple(int(d) for d in dim_y)
        self.regime = regime_of(method, do_proj)
        self.method = method
        self.n_repeats = len(repeats)
        keep = []
        arr = (Repeat * len(repeats))(*[_repeat_desc(po, tau, method, self.regime, keep)
                                       for po, tau in repeats])
        self._h = C.c_void_p()
        self._concurrency = 1
        check(self.lib.unires_plan_create(C.byref(self._h), i3(self.dim_y), f3(vx_y), self.regime,
                                          len(repeats), arr, fov_tol))
        self.dims_x = [self.dim_y if self.regime == REGIME_IDENTITY else tuple(po.dim_x)
                       for po, _ in repeats]

    def rhs_buffer(self, like):
        """A per-plan (X,Y,Z) buffer for the RHS when channels run on separate streams."""
        buf = getattr(self, '_rhs_buf', None)
        if buf is None or buf.shape != like.shape or buf.device != like.device:
            buf = torch.empty_like(like)
            self._rhs_buf = buf
        return buf

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            try:
                with torch.cuda.device(self.device):
                    self.lib.unires_plan_destroy(self._h)
            except Exception:  # interpreter shutdown
                self.lib.unires_plan_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def workspace_bytes(self):
        return int(self.lib.unires_plan_workspace_bytes(self._h))

    def repeat_info(self, n):
        """Which kernels repeat n runs on and how its voxel axes were relabelled
        (``unires_plan_repeat_info``): dict(perm, flip, pull2, splat2_axis, shift, separable, fused);
        ``fused``: the single-pass AtA kernel (denoising regime, ata1.hip) serves the CG matvec."""
        info = (C.c_int32 * 8)()
        check(self.lib.unires_plan_repeat_info(self._h, int(n), info))
        v = list(info)
        return dict(perm=tuple(v[:3]), flip=tuple((v[3] >> j) & 1 for j in range(3)), pull2=bool(v[4]),
                    splat2_axis=None if v[5] == 0 else v[5] - 2, shift=bool(v[6] & 1), separable=bool(v[7]),
                    fused=bool(v[6] & 2))

    @on_device
    def time_matvecs(self, on=True):
        """Measurement aid: bracket every operator application of the following solves with HIP events
        (the solves then run as plain launches, not as a hipGraph).  ``on=2``: no events - every A(p) is enqueued
        twice instead, inside the solve's hipGraph (same result; the time difference to a plain solve is the cost of
        the operator applications as production runs them)."""
        check(self.lib.unires_plan_time_matvecs(self._h, 2 if on == 2 else (1 if on else 0)))

    @on_device
    def matvec_time(self):
        """(launches, total microseconds) recorded since the last call; waits for them."""
        n, us = C.c_int32(0), C.c_double(0.0)
        check(self.lib.unires_plan_matvec_time(self._h, C.byref(n), C.byref(us)))
        return int(n.value), float(us.value)

    @on_device
    def set_repeat(self, n, po, tau):
        keep = []
        r = _repeat_desc(po, tau, self.method, self.regime, keep)
        check(self.lib.unires_plan_set_repeat(self._h, n, C.byref(r)))

    def set_concurrency(self, n):
        """Tell the plan that ``n`` solves run on the device at once (the channels of a y-update on streams of their
        own): its persistent kernels then leave room for the others' (``unires_plan_set_concurrency``).  Free when
        ``n`` is what the plan already has."""
        n = max(1, int(n))
        if n != self._concurrency:
            check(self.lib.unires_plan_set_concurrency(self._h, n))
            self._concurrency = n

    def _y(self, t, name):
        v, _ = _vol(t, name)
        if tuple(v.shape) != self.dim_y:
            raise ValueError('unires_amd: %s has shape %s, expected %s'
                             % (name, tuple(v.shape), self.dim_y))
        return v

    @on_device
    def proj_apply(self, n, operator, dat):
        """_proj_apply(operator, dat, po_n) without tau."""
        if operator not in _lib.OP:
            raise ValueError('Undefined operator')
        d, lead = _vol(dat)
        out_dim = self.dims_x[n] if operator == 'A' else self.dim_y
        in_dim = self.dims_x[n] if operator == 'At' else self.dim_y
        if tuple(d.shape) != tuple(in_dim):
            raise ValueError('unires_amd: input has shape %s, expected %s'
                             % (tuple(d.shape), tuple(in_dim)))
        out = torch.empty(out_dim, dtype=torch.float32, device=d.device)
        check(self.lib.unires_proj_apply(self._h, n, _lib.OP[operator], _ptr(d), _ptr(out),
                                         _stream()))
        return out.reshape(lead + tuple(out_dim))

    @on_device
    def matvec(self, p, rho, lam, out=None, dot=None):
        """q = sum tau AtA p + rho lam^2 DtD p; ``dot``: 0-d float64 device tensor or None."""
        p = self._y(p, 'p')
        if out is None:
            out = torch.empty_like(p)
        check(self.lib.unires_ata_matvec(self._h, float(rho), float(lam), _ptr(p), _ptr(out),
                                         _ptr(dot) if dot is not None else None, _stream()))
        return out

    @on_device
    def rhs(self, x_dats, w_c, z_c, rho, lam, out=None):
        """b = sum tau At x - lam Dt(w - rho z)   (unires/_update.py:124-133)."""
        xs = [_vol(t, 'x')[0] for t in x_dats]
        for t, dx in zip(xs, self.dims_x):
            if tuple(t.shape) != tuple(dx):
                raise ValueError('unires_amd: observation shape mismatch')
        if len(xs) != self.n_repeats:
            raise ValueError('unires_amd: need one observation per repeat')
        for t in (w_c, z_c):
            if tuple(t.shape) != (3,) + self.dim_y or not t.is_cuda or t.dtype != torch.float32:
                raise ValueError('unires_amd: w/z must be (3,X,Y,Z) float32 device tensors')
        w_c, z_c = w_c.contiguous(), z_c.contiguous()
        if out is None:
            out = torch.empty(self.dim_y, dtype=torch.float32, device=w_c.device)
        ptrs = (C.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
        check(self.lib.unires_rhs_assemble(self._h, ptrs, _ptr(w_c), _ptr(z_c), float(rho),
                                           float(lam), _ptr(out), _stream()))
        return out

    @on_device
    def rhs_cached(self, x_dats, w_c, z_c, rho, lam, out):
        """Same b as :meth:`rhs`, but sum_n tau_n At x_n is kept on the plan and only
        recomputed when an observation tensor changed (the plan itself is rebuilt when a
        rigid / scaling / tau changes).  The reference recomputes it every ADMM iteration
        (unires/_update.py:125-128)."""
        xs = [_vol(t, 'x')[0] for t in x_dats]
        # The cache is keyed on the caller's tensor OBJECTS (weak references) and their version
        # counters - not on addresses: a new tensor can land on a freed tensor's address with version
        # 0.  An observation that had to be copied to become contiguous is never cached.
        import weakref
        copied = any(v is not t and v.data_ptr() != t.data_ptr() for v, t in zip(xs, x_dats))
        key = None if copied else tuple((id(t), t._version) for t in x_dats)
        cache = getattr(self, '_atx', None)
        alive = cache is not None and all(r() is t for r, t in zip(cache[2], x_dats)) \
            and len(cache[2]) == len(x_dats)
        if cache is None or key is None or not alive or cache[0] != key:
            atx = cache[1] if cache is not None else torch.empty(self.dim_y, dtype=torch.float32,
                                                                 device=xs[0].device)
            ptrs = (C.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
            check(self.lib.unires_atx_assemble(self._h, ptrs, _ptr(atx), _stream()))
            refs = [] if key is None else [weakref.ref(t) for t in x_dats]
            self._atx = cache = (key, atx, refs)
        w_c, z_c = w_c.contiguous(), z_c.contiguous()
        check(self.lib.unires_rhs_from_atx(self._h, _ptr(cache[1]), _ptr(w_c), _ptr(z_c),
                                           float(rho), float(lam), _ptr(out), _stream()))
        return out

    @on_device
    def precond_build(self, rho, lam, mode='jacobi', out=None):
        """Diagonal of unires/_update.py:80-102 for this channel, kept in the plan for
        ``cg(precond='jacobi')``; ``out`` (dim_y tensor) optionally receives a copy."""
        if mode not in _lib.PRECOND:
            raise ValueError('Undefined preconditioner')
        if out is not None:
            out = self._y(out, 'out')
        check(self.lib.unires_precond_build(self._h, _lib.PRECOND[mode], float(rho), float(lam),
                                            _ptr(out) if out is not None else None, _stream()))
        return out

    @on_device
    def precond_apply(self, v, out=None):
        """out = precond(v) for the preconditioner last built on this plan (identity: copy)."""
        v = self._y(v, 'v')
        if out is None:
            out = torch.empty_like(v)
        check(self.lib.unires_precond_apply(self._h, _ptr(v), _ptr(out), _stream()))
        return out

    @on_device
    def cg(self, b, x, rho, lam, max_iter=20, tolerance=1e-3, stop='max_gain', sync=True,
           precond='none'):
        """In-place CG on x (must be contiguous (X,Y,Z)).  Returns (iters, obj) when
        ``sync`` (one stream sync), else None with everything left enqueued.
        ``precond='jacobi' | 'fft'`` needs :meth:`precond_build` with the same mode, rho, lam first.
        With a tolerance the call keeps the calling thread until the last chunk of iterations is enqueued
        (``sync=False`` does not return early then); on a capturing stream the whole solve joins the capture."""
        if precond not in _lib.PRECOND:
            raise ValueError('Undefined preconditioner')
        pm = _lib.PRECOND[precond]
        b = self._y(b, 'b')
        if not x.is_contiguous():
            raise ValueError('unires_amd: cg updates x in place and needs it contiguous')
        self._y(x, 'x')
        s = stop if stop in _lib.STOP else stop[0].lower()
        mode = _lib.STOP[s] if s in _lib.STOP else (0 if s == 'e' else 3)
        max_iter = int(max_iter)
        if sync:
            it = C.c_int32(0)
            obj = (C.c_double * (min(max_iter, 4096) + 1))()
            check(self.lib.unires_cg_solve(self._h, float(rho), float(lam), _ptr(b), _ptr(x),
                                           int(max_iter), float(tolerance), mode, pm, C.byref(it),
                                           obj, _stream()))
            trace = _trace(list(obj), it.value) if tolerance else None
            return it.value, trace
        check(self.lib.unires_cg_solve(self._h, float(rho), float(lam), _ptr(b), _ptr(x),
                                       int(max_iter), float(tolerance), mode, pm, None, None,
                                       _stream()))
        return None


def _trace(ring, it):
    """Objective trace of a solve that realised ``it`` iterations from the library's ring of ``len(ring)``
    slots (iteration k lives in slot k mod len): all it + 1 values while they fit, else the LAST len(ring) of
    them, oldest first (budgets beyond 4 096 iterations, `optim.cg(max_iter=None)`)."""
    n = len(ring)
    if it < n:
        return ring[:it + 1]
    start = (it + 1) % n
    return ring[start:] + ring[:start]


def cg_many(plans, bs, xs, rho, lams, streams, max_iter=20, tolerance=1e-3, stop='max_gain', precond='none',
            sync=False):
    """The solves of several channels at once, channel c on ``streams[c]`` (``unires_cg_solve_many``): with
    a tolerance the library feeds every solve chunk by chunk from one host loop, so the channels keep
    overlapping on the device.  Returns [(iters, obj), ...] when ``sync``."""
    if precond not in _lib.PRECOND:
        raise ValueError('Undefined preconditioner')
    n = len(plans)
    lib = plans[0].lib
    s = stop if stop in _lib.STOP else stop[0].lower()
    mode = _lib.STOP[s] if s in _lib.STOP else (0 if s == 'e' else 3)
    for pl, b, x in zip(plans, bs, xs):
        pl._y(b, 'b')
        if not x.is_contiguous():
            raise ValueError('unires_amd: cg updates x in place and needs it contiguous')
        pl._y(x, 'x')
    max_iter = int(max_iter)
    nobj = min(max_iter, 4096) + 1
    it = (C.c_int32 * n)() if sync else None
    obj = (C.c_double * (n * nobj))() if sync else None
    with torch.cuda.device(plans[0].device):
        check(lib.unires_cg_solve_many(
            n, (C.c_void_p * n)(*[pl._h.value for pl in plans]), (C.c_float * n)(*[float(rho)] * n),
            (C.c_float * n)(*[float(v) for v in lams]), (C.c_void_p * n)(*[b.data_ptr() for b in bs]),
            (C.c_void_p * n)(*[x.data_ptr() for x in xs]), max_iter, float(tolerance), mode,
            _lib.PRECOND[precond], it, obj, (C.c_void_p * n)(*[st.cuda_stream for st in streams])))
    if not sync:
        return None
    return [(it[c], _trace(list(obj[c * nobj:(c + 1) * nobj]), it[c]) if tolerance else None) for c in range(n)]
