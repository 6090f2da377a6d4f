"""Restatement (torch-CPU) of the nitorch functions the UniRes y-update calls.

ORACLE — test infrastructure only (see oracle/__init__.py).  PARITY UNPINNED:
nitorch @ 8067d60 (reference ``setup.py:11``) is absent from this container, so
everything here is a restatement of its published algorithm, cross-checked in
``tests/test_oracle.py`` against torch-native partial oracles and algebraic
properties.  Every assumed semantic that could not be checked against source is
an explicit keyword argument (``fov_tol``, ``stop`` ...), never a buried
constant.

Reference call sites (the contract each function has to honour):
  affine_grid     unires/_project.py:159
  grid_pull       unires/_project.py:164,174,183,187 ; unires/_core.py:388
  grid_push       unires/_project.py:172,179,185,188
  im_gradient     unires/_project.py:314 ; unires/_update.py:168,176,188,419
  im_divergence   unires/_project.py:315 ; unires/_update.py:132
  cg              unires/_update.py:142-148
  get_gain        unires/run.py:100
  smooth          unires/_project.py:277
  voxel_size      unires/_project.py:224,230 ; unires/_update.py:111
"""
import math

import torch

FOV_TOL = 5e-2  # nitorch's in-FOV tolerance for extrapolate=False  [recalled]


# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------
def voxel_size(mat):
    """Column norms of the 3x3 linear part of an affine (nitorch.spatial.voxel_size)."""
    mat = torch.as_tensor(mat)
    return (mat[:3, :3] ** 2).sum(0).sqrt()


def affine_grid(mat, shape):
    """Dense voxel-coordinate grid g[i,j,k,:] = lin @ (i,j,k) + off.

    nitorch.spatial.affine_grid(mat, shape, jitter=False): 'ij' meshgrid,
    0-based voxel indices, dtype of ``mat``; computed as matvec + offset.
    Returns (*shape, 3).
    """
    mat = torch.as_tensor(mat)
    dt = mat.dtype
    ax = [torch.arange(s, dtype=dt) for s in shape]
    ijk = torch.stack(torch.meshgrid(*ax, indexing='ij'), dim=-1)  # (*shape,3)
    lin = mat[:3, :3]
    off = mat[:3, 3]
    return torch.matmul(lin, ijk.unsqueeze(-1)).squeeze(-1) + off


def _corners(g, n):
    """floor index, upper index, fractional weight of the upper corner, and the
    zero-bound validity (0/1) of each of the two corners along one axis."""
    g0f = g.floor()
    w1 = g - g0f
    i0 = g0f.long()
    i1 = i0 + 1
    ok0 = ((i0 >= 0) & (i0 < n))
    ok1 = ((i1 >= 0) & (i1 < n))
    return i0.clamp(0, n - 1), i1.clamp(0, n - 1), w1, ok0, ok1


def _fov_mask(grid, shape, fov_tol):
    gx, gy, gz = grid.unbind(-1)
    nx, ny, nz = shape
    return ((gx > -fov_tol) & (gx < nx - 1 + fov_tol) &
            (gy > -fov_tol) & (gy < ny - 1 + fov_tol) &
            (gz > -fov_tol) & (gz < nz - 1 + fov_tol))


def grid_pull(inp, grid, interpolation='linear', bound='zero', extrapolate=False,
              fov_tol=FOV_TOL):
    """Trilinear gather in VOXEL coordinates, zero bound, in-FOV mask.

    inp  (B, C, X, Y, Z);  grid (B, X', Y', Z', 3) with grid[..., 0] indexing X.
    Corners outside [0, n-1] contribute zero ('zero' bound); with
    extrapolate=False the result is multiplied by the mask
    ``g_d > -tol & g_d < n_d - 1 + tol`` (tol = 5e-2).
    """
    if interpolation not in ('linear', 1, 0, 'nearest') or bound != 'zero':
        raise NotImplementedError('oracle restates linear / nearest, zero bound only')
    B, C = inp.shape[:2]
    shape = inp.shape[2:]
    nx, ny, nz = shape
    out = []
    for b in range(B):
        g = grid[b if grid.shape[0] > 1 else 0]
        gx, gy, gz = g.unbind(-1)
        if interpolation in (0, 'nearest'):
            # order 0 (the call at unires/_update.py:592-598, always at integer coordinates D u):
            # the voxel at round(g), zero outside the volume  [recalled]
            ix, iy, iz = torch.round(gx).long(), torch.round(gy).long(), torch.round(gz).long()
            ok = (ix >= 0) & (ix < nx) & (iy >= 0) & (iy < ny) & (iz >= 0) & (iz < nz)
            idx = (ix.clamp(0, nx - 1) * ny + iy.clamp(0, ny - 1)) * nz + iz.clamp(0, nz - 1)
            acc = inp[b].reshape(C, -1)[:, idx.reshape(-1)].reshape((C,) + gx.shape) * ok.to(inp.dtype)
            if not extrapolate:
                acc = acc * _fov_mask(g, shape, fov_tol).to(inp.dtype)
            out.append(acc)
            continue
        x0, x1, wx, okx0, okx1 = _corners(gx, nx)
        y0, y1, wy, oky0, oky1 = _corners(gy, ny)
        z0, z1, wz, okz0, okz1 = _corners(gz, nz)
        src = inp[b].reshape(C, -1)
        acc = torch.zeros((C,) + gx.shape, dtype=inp.dtype)
        for (ix, wxx, okx) in ((x0, 1 - wx, okx0), (x1, wx, okx1)):
            for (iy, wyy, oky) in ((y0, 1 - wy, oky0), (y1, wy, oky1)):
                for (iz, wzz, okz) in ((z0, 1 - wz, okz0), (z1, wz, okz1)):
                    idx = (ix * ny + iy) * nz + iz
                    w = (wxx * wyy * wzz) * (okx & oky & okz).to(inp.dtype)
                    acc += src[:, idx.reshape(-1)].reshape((C,) + gx.shape) * w
        if not extrapolate:
            acc = acc * _fov_mask(g, shape, fov_tol).to(inp.dtype)
        out.append(acc)
    return torch.stack(out)


def grid_grad(inp, grid, interpolation='linear', bound='zero', extrapolate=False,
              fov_tol=FOV_TOL):
    """Spatial gradient of the trilinear sample w.r.t. the voxel coordinate  [recalled]:
    (B, C, X', Y', Z', 3).  Out-of-volume corners count as zeros; with extrapolate=False the
    in-FOV mask multiplies the result (call site unires/_update.py:508)."""
    if interpolation not in ('linear', 1, 0, 'nearest') or bound != 'zero':
        raise NotImplementedError('oracle restates linear / nearest, zero bound only')
    B, C = inp.shape[:2]
    shape = inp.shape[2:]
    nx, ny, nz = shape
    out = []
    for b in range(B):
        g = grid[b if grid.shape[0] > 1 else 0]
        gx, gy, gz = g.unbind(-1)
        if interpolation in (0, 'nearest'):
            # order 0 (the call at unires/_update.py:592-598, always at integer coordinates D u):
            # the voxel at round(g), zero outside the volume  [recalled]
            ix, iy, iz = torch.round(gx).long(), torch.round(gy).long(), torch.round(gz).long()
            ok = (ix >= 0) & (ix < nx) & (iy >= 0) & (iy < ny) & (iz >= 0) & (iz < nz)
            idx = (ix.clamp(0, nx - 1) * ny + iy.clamp(0, ny - 1)) * nz + iz.clamp(0, nz - 1)
            acc = inp[b].reshape(C, -1)[:, idx.reshape(-1)].reshape((C,) + gx.shape) * ok.to(inp.dtype)
            if not extrapolate:
                acc = acc * _fov_mask(g, shape, fov_tol).to(inp.dtype)
            out.append(acc)
            continue
        x0, x1, wx, okx0, okx1 = _corners(gx, nx)
        y0, y1, wy, oky0, oky1 = _corners(gy, ny)
        z0, z1, wz, okz0, okz1 = _corners(gz, nz)
        src = inp[b].reshape(C, -1)
        acc = torch.zeros((C,) + gx.shape + (3,), dtype=inp.dtype)
        one = torch.ones_like(wx)
        for (ix, wxx, dxx, okx) in ((x0, 1 - wx, -one, okx0), (x1, wx, one, okx1)):
            for (iy, wyy, dyy, oky) in ((y0, 1 - wy, -one, oky0), (y1, wy, one, oky1)):
                for (iz, wzz, dzz, okz) in ((z0, 1 - wz, -one, okz0), (z1, wz, one, okz1)):
                    idx = (ix * ny + iy) * nz + iz
                    ok = (okx & oky & okz).to(inp.dtype)
                    v = src[:, idx.reshape(-1)].reshape((C,) + gx.shape) * ok
                    acc[..., 0] += v * (dxx * wyy * wzz)
                    acc[..., 1] += v * (wxx * dyy * wzz)
                    acc[..., 2] += v * (wxx * wyy * dzz)
        if not extrapolate:
            acc = acc * _fov_mask(g, shape, fov_tol).to(inp.dtype)[..., None]
        out.append(acc)
    return torch.stack(out)


def grid_push(inp, grid, shape, interpolation='linear', bound='zero', extrapolate=False,
              fov_tol=FOV_TOL):
    """Exact adjoint of grid_pull w.r.t. its input (scatter-add of 8 corners).

    inp (B, C, X', Y', Z');  grid (B, X', Y', Z', 3);  returns (B, C, *shape).
    """
    if interpolation not in ('linear', 1) or bound != 'zero':
        raise NotImplementedError('oracle restates linear/zero only')
    B, C = inp.shape[:2]
    nx, ny, nz = shape
    out = torch.zeros((B, C, nx * ny * nz), dtype=inp.dtype)
    for b in range(B):
        g = grid[b if grid.shape[0] > 1 else 0]
        gx, gy, gz = g.unbind(-1)
        x0, x1, wx, okx0, okx1 = _corners(gx, nx)
        y0, y1, wy, oky0, oky1 = _corners(gy, ny)
        z0, z1, wz, okz0, okz1 = _corners(gz, nz)
        val = inp[b].reshape(C, -1)
        if not extrapolate:
            val = val * _fov_mask(g, shape, fov_tol).reshape(-1).to(inp.dtype)
        for (ix, wxx, okx) in ((x0, 1 - wx, okx0), (x1, wx, okx1)):
            for (iy, wyy, oky) in ((y0, 1 - wy, oky0), (y1, wy, oky1)):
                for (iz, wzz, okz) in ((z0, 1 - wz, okz0), (z1, wz, okz1)):
                    idx = ((ix * ny + iy) * nz + iz).reshape(-1)
                    w = ((wxx * wyy * wzz) * (okx & oky & okz).to(inp.dtype)).reshape(-1)
                    out[b].index_add_(1, idx, val * w)
    return out.reshape(B, C, nx, ny, nz)


# --------------------------------------------------------------------------
# finite differences (forward, zero bound)
# --------------------------------------------------------------------------
def im_gradient(dat, vx=None, which='forward', bound='zero'):
    """Forward differences with zero bound: g_d[i] = (y[i+e_d] - y[i]) / vx_d,
    y[n_d] := 0.  (X,Y,Z) -> (3,X,Y,Z)."""
    if which != 'forward' or bound != 'zero':
        raise NotImplementedError('oracle restates forward/zero only')
    vx = torch.ones(3) if vx is None else torch.as_tensor(vx, dtype=dat.dtype)
    out = []
    for d in range(3):
        nxt = torch.zeros_like(dat)
        sl_to = [slice(None)] * 3
        sl_from = [slice(None)] * 3
        sl_to[d] = slice(0, -1)
        sl_from[d] = slice(1, None)
        nxt[tuple(sl_to)] = dat[tuple(sl_from)]
        out.append((nxt - dat) / vx[d])
    return torch.stack(out)


def im_divergence(dat, vx=None, which='forward', bound='zero'):
    """POSITIVE adjoint of im_gradient (no minus sign):
    (D^T g)[i] = sum_d (g_d[i-e_d] - g_d[i]) / vx_d,  g_d[-1] := 0.
    (3,X,Y,Z) -> (X,Y,Z)."""
    if which != 'forward' or bound != 'zero':
        raise NotImplementedError('oracle restates forward/zero only')
    vx = torch.ones(3) if vx is None else torch.as_tensor(vx, dtype=dat.dtype)
    out = torch.zeros_like(dat[0])
    for d in range(3):
        g = dat[d]
        prv = torch.zeros_like(g)
        sl_to = [slice(None)] * 3
        sl_from = [slice(None)] * 3
        sl_to[d] = slice(1, None)
        sl_from[d] = slice(0, -1)
        prv[tuple(sl_to)] = g[tuple(sl_from)]
        out = out + (prv - g) / vx[d]
    return out


# --------------------------------------------------------------------------
# optimisation
# --------------------------------------------------------------------------
def get_gain(obj, monotonicity='increasing'):
    """(obj[-2]-obj[-1]) / (max(obj)-min(obj)) for 'decreasing'; inf if len<=1.
    Pinned by the reference's printed trace
    (demos/demo_single_channel.ipynb:173-175: inf, 1.0, 0.3567)."""
    if len(obj) <= 1:
        return torch.tensor(float('inf'), dtype=torch.float64)
    if monotonicity == 'increasing':
        gain = obj[-1] - obj[-2]
    else:
        gain = obj[-2] - obj[-1]
    return gain / (torch.max(obj) - torch.min(obj))


def cg(A, b, x=None, precond=lambda y: y, max_iter=None, tolerance=1e-5,
       verbose=False, sum_dtype=torch.float64, inplace=True, stop='E',
       return_info=False):
    """(Preconditioned) conjugate gradients, as nitorch.core.optim.cg  [recalled].

    Convergence objective: ``stop[0].lower() == 'e'`` -> sqrt(r.z); any other
    first letter ('max_gain' -> 'm', which is what UniRes passes,
    unires/_update.py:145) -> 0.5 * sum(x * (A(x) - 2 b)), i.e. ONE EXTRA A(x)
    per iteration.  Stop when |get_gain(obj[:k+1], 'decreasing')| < tolerance.
    alpha/beta are 0-d float64 tensors; ``alpha * p`` with float32 p stays float32.
    """
    if max_iter is None:
        max_iter = b.numel() * 10
    if x is None:
        x = torch.zeros_like(b)
    elif not inplace:
        x = x.clone()

    r = b - A(x)
    z = precond(r)
    rz = torch.sum(r * z, dtype=sum_dtype)
    p = z.clone()

    check = bool(tolerance) or verbose
    if check:
        if stop == 'residual':
            stop = 'e'
        elif stop == 'norm':
            stop = 'a'
        stop = stop[0].lower()
        if stop == 'e':
            obj0 = torch.sqrt(rz)
        else:
            obj0 = 0.5 * torch.sum(A(x).sub_(2 * b).mul_(x), dtype=sum_dtype)
        obj = torch.zeros(max_iter + 1, dtype=sum_dtype)
        obj[0] = obj0

    n_done = 0
    for n_iter in range(1, max_iter + 1):
        Ap = A(p)
        alpha = rz / torch.sum(p * Ap, dtype=sum_dtype)
        x += alpha * p
        r -= alpha * Ap
        z = precond(r)
        rz0 = rz
        rz = torch.sum(r * z, dtype=sum_dtype)
        beta = rz / rz0
        p *= beta
        p += z
        n_done = n_iter
        if check:
            if stop == 'e':
                obj1 = torch.sqrt(rz)
            else:
                obj1 = 0.5 * torch.sum(A(x).sub_(2 * b).mul_(x), dtype=sum_dtype)
            obj[n_iter] = obj1
            gain = get_gain(obj[:n_iter + 1], monotonicity='decreasing')
            if verbose:
                print('{:3d} | {} = {:12.6g} | gain = {:12.6g}'.format(n_iter, stop, obj1, gain))
            if gain.abs() < tolerance:
                break
    if return_info:
        return x, n_done, (obj[:n_done + 1].clone() if check else None)
    return x


# --------------------------------------------------------------------------
# slice-profile kernels
# --------------------------------------------------------------------------
def _gl_integrate(f, lo, hi, breaks):
    """Exact-for-piecewise-quintic integration of f on [lo,hi] (3-pt Gauss-Legendre
    on every sub-interval delimited by ``breaks``)."""
    pts = sorted(set([lo, hi] + [t for t in breaks if lo < t < hi]))
    nodes = (-math.sqrt(3.0 / 5.0), 0.0, math.sqrt(3.0 / 5.0))
    wts = (5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0)
    tot = 0.0
    for a, b in zip(pts[:-1], pts[1:]):
        h, m = 0.5 * (b - a), 0.5 * (a + b)
        tot += h * sum(w * f(m + h * n) for n, w in zip(nodes, wts))
    return tot


def _tri(t):
    return max(0.0, 1.0 - abs(t))


def smooth1d(kind, fwhm, gauss_lim=None):
    """1-D slice profile convolved with the linear-interpolation basis, sampled
    at integer offsets and normalised to sum 1  (nitorch.core.kernels.smooth,
    basis=1)  [recalled].

    kind: -1 dirac | 0 rect | 1 tri | 2 gauss.  Support x in [-L, L]:
      rect  L = floor((w+2)/2)     (w=4 -> [0,.125,.25,.25,.25,.125,0])
      tri   L = floor((2w+2)/2)
      gauss L = floor((4w+2)/2)    -- truncation UNPINNED; override via gauss_lim.
            Two recollections of nitorch's `_gauss1` disagree: this one (w = FWHM: 11 taps
            at ratio 2) and `lim = floor(4 sigma + 1)` with sigma = w / sqrt(8 ln 2) (9 taps at ratio 2).  The
            two outer taps are ~1e-8 of the sum, so A differs by float32 rounding only; what WOULD differ is
            `po.smo_ker.shape` / `po.dim_yx` (and `mat_yx`'s offset, which compensates: unires/_project.py:280-285).
            Kept at 11 until someone runs tests/golden/make_golden_from_reference.py --real-nitorch; every
            fixture records the kernel it was made with (`po_smo_ker`), so a regenerated set shows the answer.
    Zero end-taps do not change the operator A: the offset compensation in
    unires/_project.py:280-285 makes A invariant to symmetric zero padding.
    """
    w = float(fwhm)
    if kind == -1:
        return [1.0]
    if kind == 0:
        L = int(math.floor((w + 2) / 2))
        ker = [_gl_integrate(_tri, x - w / 2, x + w / 2, [-1.0, 0.0, 1.0]) / w
               for x in range(-L, L + 1)]
    elif kind == 1:
        L = int(math.floor((2 * w + 2) / 2))
        ker = []
        for x in range(-L, L + 1):
            f = lambda t, x=x: (_tri(t / w) / w) * _tri(x - t)
            ker.append(_gl_integrate(f, -w, w, [0.0, x - 1.0, float(x), x + 1.0]))
    elif kind == 2:
        L = int(math.floor((4 * w + 2) / 2)) if gauss_lim is None else int(gauss_lim)
        s = (w / math.sqrt(8.0 * math.log(2.0))) ** 2 + 1e-12
        w1 = 0.5 * math.sqrt(2.0 / s)
        w2 = -0.5 / s
        w3 = math.sqrt(s / (2.0 * math.pi))
        ker = []
        for x in range(-L, L + 1):
            k = 0.5 * (math.erf(w1 * (x + 1)) * (x + 1) + math.erf(w1 * (x - 1)) * (x - 1)
                       - 2 * math.erf(w1 * x) * x) \
                + w3 * (math.exp(w2 * (x + 1) ** 2) + math.exp(w2 * (x - 1) ** 2)
                        - 2 * math.exp(w2 * x ** 2))
            ker.append(max(k, 0.0))
    else:
        raise ValueError('unknown profile')
    tot = sum(ker)
    return [k / tot for k in ker]


def smooth(types, fwhm, sep=False, dtype=torch.float32, gauss_lim=None):
    """Outer product of per-axis smooth1d kernels -> (1, 1, kx, ky, kz)
    (the ``sep=False`` form used at unires/_project.py:277)."""
    k1 = [torch.tensor(smooth1d(int(t), float(f), gauss_lim), dtype=torch.float64)
          for t, f in zip(types, fwhm)]
    if sep:
        return [k.to(dtype) for k in k1]
    ker = k1[0][:, None, None] * k1[1][None, :, None] * k1[2][None, None, :]
    return ker.to(dtype)[None, None]
