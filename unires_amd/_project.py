"""Projection operators of the y-update - host-side mirror of
unires/_project.py (same names, argument meaning and errors), backed by HIP.

    _apply_scaling   unires/_project.py:9-24
    _check_adjoint   unires/_project.py:27-51
    _proj            unires/_project.py:54-96
    _proj_apply      unires/_project.py:99-190
    _proj_info       unires/_project.py:193-297
    _DtD             unires/_project.py:300-317
"""
import math

import numpy as np
import torch

from . import _kernels, _ops
from ._plan import ChannelPlan, proj_matrix
from .spatial import _m12, voxel_size
from .struct import _proj_op

_F64 = torch.float64


def _apply_scaling(dat, scl, dim):
    """Even/odd slice scaling along ``dim`` (a conv with a 1-tap identity kernel and
    the scaling epilogue of the conv kernel)."""
    one = [np.ones(1, np.float32)] * 3
    return _ops.conv_down(dat, one, (1, 1, 1), scl=float(scl), scl_dim=int(dim))


def _proj_info(dim_y, mat_y, dim_x, mat_x, rigid=None, prof_ip=0, prof_tp=0, gap=0.0,
               device='cuda', scl=0.0, samp=0, gauss_lim=None):
    """Define a projection operator object for _proj_apply.  All 4x4 arithmetic is
    float64 on the host (it is set-up work: once per input, again per rigid update)."""
    po = _proj_op()
    mat_y = torch.as_tensor(mat_y).detach().to('cpu', _F64)
    mat_x = torch.as_tensor(mat_x).detach().to('cpu', _F64)
    dim_y = tuple(int(d) for d in dim_y)
    dim_x = tuple(int(d) for d in dim_x)
    if len(dim_y) != 3 or len(dim_x) != 3:
        raise ValueError('only 3-D volumes are built')
    po.dim_y, po.mat_y, po.vx_y = dim_y, mat_y, voxel_size(mat_y)
    po.dim_x, po.mat_x, po.vx_x = dim_x, mat_x, voxel_size(mat_x)
    po.rigid = torch.eye(4, dtype=_F64) if rigid is None \
        else torch.as_tensor(rigid).detach().to('cpu', _F64)
    po.D_x = po.D_y = None
    # thick-slice axis and per-axis profile / gap: from the ORIGINAL voxel size, before any
    # sub-sampling (unires/_project.py:239-243 sit above the samp block :245-264; decimation can move
    # the argmax: vx_x = (0.4, 0.4, 1), samp = 1 decimates to (0.8, 0.8, 1) - still z - but
    # vx_x = (0.5, 0.5, 0.8) becomes (1, 1, 0.8))
    po.dim_thick = int(torch.max(po.vx_x, dim=0)[1])
    profile = [int(prof_ip)] * 3
    gaps = [0.0] * 3
    profile[po.dim_thick] = int(prof_tp)
    gaps[po.dim_thick] = float(gap)
    if samp > 0:
        # sub-sampling for the rigid Gauss-Newton (unires/_project.py:245-264): the low-res image is
        # decimated by sk = max(1, floor(samp / vx_x + 0.5)) voxels per axis: mat_x <- mat_x D_x,
        # dim_x <- floor(dim_x / sk).  (The high-res branch is dead code in the reference - :255
        # compares vx_x with itself - so D_y stays None.)  po.sk is the decimation the caller
        # applies to the data: nearest-neighbour pull at the integer coordinates D_x u = a strided
        # slice (unires/_update.py:589-593).
        sk = torch.clamp(torch.floor(float(samp) / po.vx_x + 0.5), min=1.0)
        po.sk = tuple(int(v) for v in sk.tolist())
        po.D_x = torch.diag(torch.cat((sk, torch.ones(1, dtype=_F64))))
        mat_x = mat_x.mm(po.D_x)
        dim_x = tuple(int(math.floor(d / k)) for d, k in zip(dim_x, po.sk))
        if min(dim_x) < 1:
            raise ValueError('sub-sampling leaves an empty image')
        po.dim_x, po.mat_x, po.vx_x = dim_x, mat_x, voxel_size(mat_x)
    # low-res / high-res voxel ratio, rounded up, at least one
    lin = torch.linalg.solve(mat_y, mat_x)[:3, :3]
    ratio = (lin ** 2).sum(0).sqrt().ceil().clamp(1)
    po.ratio = tuple(int(r) for r in ratio.tolist())
    # intermediate (high-res sampling of the low-res FOV) space
    po.mat_yx = mat_x.matmul(torch.diag(torch.cat((1.0 / ratio, torch.ones(1, dtype=_F64)))))
    dim_yx = [(dx - 1) * r + 1 for dx, r in zip(dim_x, po.ratio)]
    # slice profile (dirac where the ratio is one), separable factors + dense form
    k1 = []
    for d in range(3):
        kind = -1 if po.ratio[d] == 1 else profile[d]
        k1.append(_kernels.smooth1d(kind, (1.0 - gaps[d]) * po.ratio[d], gauss_lim))
    po.smo_ker_1d = [torch.from_numpy(k.astype(np.float32)) for k in k1]
    dense = k1[0][:, None, None] * k1[1][None, :, None] * k1[2][None, None, :]
    po.smo_ker = torch.from_numpy(dense.astype(np.float32))[None, None].to(device)
    # centre the kernel: shift the intermediate space by floor(-(k-1)/2) and grow it
    off = [int(math.floor(-(len(k) - 1) / 2.0)) for k in k1]
    mat_off = torch.eye(4, dtype=_F64)
    mat_off[:3, 3] = torch.tensor(off, dtype=_F64)
    po.mat_yx = po.mat_yx.matmul(mat_off)
    po.dim_yx = tuple(int(n + 2 * abs(o)) for n, o in zip(dim_yx, off))
    po.scl = scl if isinstance(scl, torch.Tensor) else torch.tensor(scl, dtype=torch.float32)
    return po


def _taps(po):
    if getattr(po, 'smo_ker_1d', None) is None:  # user supplied only the dense kernel
        po.smo_ker_1d = [torch.from_numpy(k) for k in
                         _kernels.factorise(po.smo_ker.detach().cpu().numpy())]
    return [k.detach().cpu().numpy() for k in po.smo_ker_1d]


def _proj_apply(operator, dat, po, method='super-resolution', bound='zero',
                interpolation='linear'):
    """Applies A, At or AtA (denoising or super-resolution) to (1,1,X,Y,Z) data.
    Composed from op-level HIP kernels with on-the-fly coordinates."""
    if operator not in ['A', 'At', 'AtA', 'none']:
        raise ValueError('Undefined operator')
    if method not in ['denoising', 'super-resolution']:
        raise ValueError('Undefined method')
    if bound != 'zero' or interpolation not in ('linear', 1):
        raise NotImplementedError("only bound='zero', interpolation='linear' are built")
    if operator == 'none':
        return dat
    mat, dim_g = proj_matrix(po, method)
    M = _m12(mat)
    scl = float(po.scl)
    if method == 'super-resolution':
        taps = _taps(po)
        if operator == 'A':
            return _ops.conv_down(_ops.pull_affine(dat, M, dim_g), taps, po.ratio, scl,
                                  po.dim_thick)
        if operator == 'At':
            return _ops.push_affine(_ops.conv_up(dat, taps, po.ratio, scl, po.dim_thick), M,
                                    po.dim_y)
        low = _ops.conv_down(_ops.pull_affine(dat, M, dim_g), taps, po.ratio, 2.0 * scl,
                             po.dim_thick)
        return _ops.push_affine(_ops.conv_up(low, taps, po.ratio), M, po.dim_y)
    if operator == 'A':
        return _ops.pull_affine(dat, M, dim_g)
    if operator == 'At':
        return _ops.push_affine(dat, M, po.dim_y)
    return _ops.push_affine(_ops.pull_affine(dat, M, dim_g), M, po.dim_y)


def _plan_signature(x, y, method, do):
    sig = [method, bool(do), tuple(y.dim)]
    for xn in x:
        po = xn.po
        if do:
            mat, dim_g = proj_matrix(po, method)
            taps = tuple(tuple(float(v) for v in k) for k in _taps(po)) \
                if method == 'super-resolution' else ()
            sig.append((tuple(_m12(mat).tolist()), dim_g, tuple(po.dim_x), float(po.scl),
                        float(xn.tau), tuple(po.ratio), taps))
        else:
            sig.append((float(xn.tau),))
    return tuple(sig)


def _channel_plan(x, y, method, do, vx_y=None):
    """Fused per-channel plan, cached on the output struct and rebuilt when any
    operator parameter (rigid, scl, tau, dims) changed."""
    sig = _plan_signature(x, y, method, do)
    cached = getattr(y, '_plan', None)
    if cached is not None and cached[0] == sig:
        return cached[1]
    if cached is not None and len(cached[0]) == len(sig) and cached[0][:3] == sig[:3]:
        # same volume and method, some repeat's operator changed (rigid / scaling update):
        # swap the descriptors in place, keeping the plan's device workspace
        plan = cached[1]
        try:
            for n, (old, new) in enumerate(zip(cached[0][3:], sig[3:])):
                if old != new:
                    plan.set_repeat(n, x[n].po, x[n].tau)
            plan.dims_x = [tuple(xn.po.dim_x) for xn in x] if do else plan.dims_x
            plan._atx = None  # cached sum tau At x belongs to the old operator
            y._plan = (sig, plan)
            return plan
        except ValueError:
            pass  # the new repeat does not fit the workspace: build a new plan
    if cached is not None:
        cached[1].close()
    if vx_y is None:
        vx_y = voxel_size(y.mat)
    vx = [float(v) for v in torch.as_tensor(vx_y).detach().cpu().tolist()]
    plan = ChannelPlan(y.dim, vx, [(xn.po, xn.tau) for xn in x], method, do, device=y.dat.device)
    y._plan = (sig, plan)
    return plan


def _proj(operator, dat, x, y, method='super-resolution', do=True, rho=1, n=0, vx_y=None,
          interpolation='linear', bound='zero', diff='forward'):
    """Projects image data by A, At or AtA; ``x`` is the list of repeats of one
    channel, ``y`` its output struct.  'AtA' is the fused
    sum_n tau_n AtA_n dat + rho lam^2 DtD dat."""
    if bound != 'zero' or diff != 'forward' or interpolation not in ('linear', 1):
        raise NotImplementedError("only bound='zero', diff='forward', linear are built")
    plan = _channel_plan(x, y, method, do, vx_y)
    if operator == 'AtA':
        return plan.matvec(dat, float(rho), float(y.lam))
    if operator not in ('A', 'At'):
        raise ValueError('Undefined operator')
    return plan.proj_apply(n, operator, dat)


def _DtD(dat, vx_y, bound='zero', diff='forward'):
    """Divergence of the gradient, one 7-point stencil pass."""
    if bound != 'zero' or diff != 'forward':
        raise NotImplementedError("only bound='zero', diff='forward' are built")
    return _ops.dtd(dat, vx_y, a=0.0, c=1.0)


def _check_adjoint(po, method, bound='zero', interpolation='linear', dtype=torch.float32):
    """<Ay, x> - <Atx, y> with seed-0 uniform inputs.  Returns the value (the
    reference prints it).  The kernels are float32; float64 is not built."""
    if dtype != torch.float32:
        raise NotImplementedError('the HIP kernels are float32')
    torch.manual_seed(0)
    dev = po.smo_ker.device
    x = torch.rand((1, 1) + tuple(po.dim_x), dtype=dtype).to(dev)
    y = torch.rand((1, 1) + tuple(po.dim_y), dtype=dtype).to(dev)
    Ay = _proj_apply('A', y, po, method=method, bound=bound, interpolation=interpolation)
    Atx = _proj_apply('At', x, po, method=method, bound=bound, interpolation=interpolation)
    val = torch.sum(Ay * x, dtype=torch.float64) - torch.sum(Atx * y, dtype=torch.float64)
    return val.item()
