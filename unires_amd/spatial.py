"""nitorch.spatial-shaped entry points for the functions UniRes imports from it
(unires/_project.py:2-3, unires/_update.py:5-7), backed by the HIP library.

The dense coordinate grid that the reference materialises on every operator call
(affine_grid, unires/_project.py:159) never exists here: pull/push take the 4x4
affine itself and compute coordinates in registers.
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


def grid_pull(input, mat, shape, interpolation='linear', bound='zero', extrapolate=False):
    """nitorch grid_pull(input, affine_grid(mat, shape), ...)."""
    _only_linear_zero(interpolation, bound, extrapolate)
    return _ops.pull_affine(input, _m12(mat), shape)


def grid_grad(input, mat, shape, interpolation='linear', bound='zero', extrapolate=False):
    """nitorch grid_grad(input, affine_grid(mat, shape), ...) -> (..., *shape, 3)."""
    _only_linear_zero(interpolation, bound, extrapolate)
    return _ops.pull_grad_affine(input, _m12(mat), shape)


def grid_push(input, mat, shape, interpolation='linear', bound='zero', extrapolate=False):
    """nitorch grid_push(input, affine_grid(mat, input.shape[-3:]), shape=shape, ...)."""
    _only_linear_zero(interpolation, bound, extrapolate)
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
