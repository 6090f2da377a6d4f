"""nitorch.spatial-shaped entry points for the functions UniRes imports from it
(unires/_project.py:2-3, unires/_update.py:5-7), backed by the HIP library.

The dense coordinate grid that the reference materialises on every operator call
(affine_grid, unires/_project.py:159) never exists here: pull/push take the 4x4
affine itself and compute coordinates in registers.  Two calling forms are accepted:

    grid_pull(input, mat, shape)        # the efficient one: the affine and the grid shape
    grid_pull(input, grid)              # nitorch's own: a dense (1, X, Y, Z, 3) grid that IS an
                                        # affine grid (what unires/_project.py:159 builds); the
                                        # affine is recovered from it and checked.  A grid that is
                                        # not affine raises NotImplementedError: the path only ever
                                        # builds affine grids.
This is a convenience layer over ``_ops``; the drop-in seam of the path is one level up
(``_proj`` / ``_proj_apply`` / ``_update_admm``).
"""
import numpy as np
import torch

from . import _ops


def voxel_size(mat):
    """Column norms of the linear part of an affine (float64 CPU tensor)."""
    m = torch.as_tensor(mat, dtype=torch.float64, device='cpu')
    return (m[:3, :3] ** 2).sum(0).sqrt()


def _m12(mat):
    """float32 row-major 3x4 of a (4,4)/(3,4) affine - the cast the reference does
    at grid creation (mat.type(dat.dtype), unires/_project.py:159)."""
    m = np.asarray(torch.as_tensor(mat).detach().cpu().numpy(), dtype=np.float64)
    return m[:3, :4].astype(np.float32).reshape(-1)


def affine_grid(mat, shape, jitter=False):
    """Dense voxel-coordinate grid ``(*shape, 3)`` of an affine, float32 like the reference's call
    (unires/_project.py:159).  Only for callers that insist on a grid: the kernels never read one."""
    m = torch.as_tensor(mat).detach().cpu().to(torch.float32)
    ax = [torch.arange(int(n), dtype=torch.float32) for n in shape]
    ijk = torch.stack(torch.meshgrid(*ax, indexing='ij'), -1)
    return ijk @ m[:3, :3].T + m[:3, 3]


def _affine_of_grid(grid):
    """(mat 4x4 float64, shape) of a dense grid that is an affine grid, else NotImplementedError."""
    g = torch.as_tensor(grid).detach()
    if g.dim() == 5 and g.shape[0] == 1:
        g = g[0]
    if g.dim() != 4 or g.shape[-1] != 3:
        raise ValueError('grid must be (1, X, Y, Z, 3) or (X, Y, Z, 3)')
    shape = tuple(int(n) for n in g.shape[:3])
    g = g.to('cpu', torch.float64)
    o = g[0, 0, 0]
    mat = torch.eye(4, dtype=torch.float64)
    mat[:3, 3] = o
    for d in range(3):
        if shape[d] > 1:
            idx = [0, 0, 0]
            idx[d] = shape[d] - 1
            mat[:3, d] = (g[tuple(idx)] - o) / (shape[d] - 1)
    # EVERY grid point must lie on the affine (float32 grids: a few ulps of the coordinates): a dense
    # deformation that happens to vanish at a handful of probe points must not be sampled as if it were
    # one (one vectorised pass on the host; this facade is a convenience path, not the hot one)
    tol = 1e-4 * max(1.0, float(g.abs().max()))
    ax = [torch.arange(n, dtype=torch.float64) for n in shape]
    ijk = torch.stack(torch.meshgrid(*ax, indexing='ij'), -1)
    want = ijk @ mat[:3, :3].T + mat[:3, 3]
    if float((g - want).abs().max()) > tol:
        raise NotImplementedError('unires_amd: only affine sampling grids are built (the reference '
                                  'path never makes another kind, unires/_project.py:159)')
    return mat, shape


def _mat_shape(mat_or_grid, shape):
    t = torch.as_tensor(mat_or_grid)
    if t.dim() >= 4:
        return _affine_of_grid(t)
    if shape is None:
        raise ValueError('shape is required with an affine matrix')
    return t, tuple(int(n) for n in shape)


def grid_pull(input, mat, shape=None, interpolation='linear', bound='zero', extrapolate=False):
    """nitorch grid_pull(input, affine_grid(mat, shape), ...); ``mat`` may be that grid itself."""
    _only_linear_zero(interpolation, bound, extrapolate)
    m, shp = _mat_shape(mat, shape)
    return _ops.pull_affine(input, _m12(m), shp)


def grid_grad(input, mat, shape=None, interpolation='linear', bound='zero', extrapolate=False):
    """nitorch grid_grad(input, affine_grid(mat, shape), ...) -> (..., *shape, 3)."""
    _only_linear_zero(interpolation, bound, extrapolate)
    m, shp = _mat_shape(mat, shape)
    return _ops.pull_grad_affine(input, _m12(m), shp)


def grid_push(input, mat, shape, interpolation='linear', bound='zero', extrapolate=False):
    """nitorch grid_push(input, affine_grid(mat, input.shape[-3:]), shape=shape, ...); ``mat`` may
    be that grid itself (its spatial shape must be the input's)."""
    _only_linear_zero(interpolation, bound, extrapolate)
    t = torch.as_tensor(mat)
    if t.dim() >= 4:
        m, gshape = _affine_of_grid(t)
        if tuple(input.shape[-3:]) != gshape:
            raise ValueError('grid_push: grid and input shapes differ')
        mat = m
    return _ops.push_affine(input, _m12(mat), shape)


def im_gradient(dat, vx=None, which='forward', bound='zero'):
    _only_forward_zero(which, bound)
    return _ops.grad_fwd_zero(dat, vx)


def im_divergence(dat, vx=None, which='forward', bound='zero'):
    _only_forward_zero(which, bound)
    return _ops.div_fwd_zero(dat, vx)


def _only_linear_zero(interpolation, bound, extrapolate):
    if interpolation not in ('linear', 1) or bound != 'zero' or extrapolate:
        raise NotImplementedError('unires_amd builds the reference defaults only: '
                                  "interpolation='linear', bound='zero', extrapolate=False "
                                  '(unires/struct.py:64,85; unires/_project.py:162,181)')


def _only_forward_zero(which, bound):
    if which != 'forward' or bound != 'zero':
        raise NotImplementedError("unires_amd builds the reference defaults only: "
                                  "diff='forward', bound='zero' (unires/struct.py:64,74)")
