"""Where the HIP path and the oracle may legitimately differ - and nowhere else.

The reference's in-FOV mask (nitorch grid_pull / grid_push with extrapolate=False: a sample counts
iff every coordinate g satisfies -5e-2 < g < n - 1 + 5e-2; call sites unires/_project.py:164-188)
is DISCONTINUOUS at the thresholds: which side a float32 coordinate falls on depends on the
last-ulp rounding of the coordinate arithmetic (torch-CPU matmul in the reference, an FMA chain in
the kernels).  The other parity tests draw their geometries away from such ties
(tests/helpers.py: fov_margin > 1e-4).  This one SEEKS them - rigid transforms whose grid planes
sit on the thresholds - and asserts the converse the other tests leave open: every disagreement
beyond float32 round-off lies within reach of a grid point whose coordinate is within 1e-4 of a
threshold; everywhere else the two agree to 1e-5.
"""
import math

import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import SIGNED_PERMS, orient_axes, rigid_matrix
from unires_amd._plan import ChannelPlan

pytestmark = pytest.mark.gpu

TOL_MASK = 5e-2


def _tie_points(mat, dim_g, dim_y, eps):
    """Grid points with a coordinate within eps of an in-FOV threshold (bool volume over the grid)."""
    g = N.affine_grid(mat.float(), dim_g)
    near = torch.zeros(g.shape[:3], dtype=torch.bool)
    for d, n in enumerate(dim_y):
        for thr in (-TOL_MASK, n - 1 + TOL_MASK):
            near |= (g[..., d] - thr).abs() < eps
    return near, g


def _reach(near, g, dim_y, reach):
    """Output voxels a tie point's trilinear footprint can touch."""
    out = torch.zeros(dim_y, dtype=torch.bool)
    for pt in g[near]:
        lo = [int(max(0, math.floor(float(v)) - reach + 1)) for v in pt]
        hi = [int(min(n, math.floor(float(v)) + reach + 1)) for v, n in zip(pt, dim_y)]
        if all(h > l for l, h in zip(lo, hi)):
            out[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
    return out


# translations that put the first / last grid planes ON the thresholds (-0.05 and n - 1 + 0.05 for a
# grid of the volume's own size), with rotations small enough that whole rows of grid points stay
# within 1e-4 of them and large enough that the coordinates are no longer exact in float32
GEOMS = [((-0.05, 0.05, -0.05), (2e-6, -3e-6, 1e-6)),
         ((0.05, -0.05, 0.05), (-2e-6, 1e-6, 3e-6)),
         ((-0.05, -0.05, 0.05), (4e-6, 4e-6, -2e-6)),
         ((-0.0500001, 0.0499999, -0.05), (0.0, 0.0, 0.0)),
         ((0.3, -0.05, 0.2), (1e-5, 0.0, -1e-5))]


@pytest.mark.parametrize('trans,rot', GEOMS)
def test_denoising_disagreements_lie_within_reach_of_fov_ties(dev, trans, rot):
    import unires_amd as U
    dim_y = (26, 24, 22)
    mat_y = torch.eye(4, dtype=torch.float64)
    rigid = rigid_matrix(list(trans), list(rot))
    po_o = O.proj_info(dim_y, mat_y, dim_y, mat_y, rigid=rigid)
    po_g = U._proj_info(dim_y, mat_y, dim_y, mat_y, rigid=rigid, device=dev)
    mat, dim_g = O.proj_matrix(po_o, 'denoising')
    near, g = _tie_points(mat, dim_g, dim_y, eps=1e-4)
    assert int(near.sum()) > 0, 'the geometry was meant to put grid points on the thresholds'
    torch.manual_seed(5)
    p = torch.rand(dim_y) + 0.5
    v = torch.rand(dim_g) + 0.5
    cover = _reach(near, g, dim_y, reach=2)
    # both sets of kernels: the plan's (LDS-window pull, schedule-driven splat - the hot path) and the
    # op-level ones U._proj_apply composes
    plan = ChannelPlan(dim_y, (1.0, 1.0, 1.0), [(po_g, 1.0)], 'denoising', True, device=dev)
    kernels = {'plan': lambda op, a: plan.proj_apply(0, op, a.to(dev)).cpu(),
               'op-level': lambda op, a: U._proj_apply(op, a[None, None].to(dev), po_g, method='denoising')[0, 0].cpu()}
    for name, apply in kernels.items():
        # A: a flipped mask changes exactly that grid point
        ref = O.proj_apply('A', p[None, None], po_o, method='denoising')[0, 0]
        out = apply('A', p)
        bad = (out - ref).abs() > 1e-5 * float(ref.abs().max())
        assert not bool((bad & ~near).any()), 'A (%s) differs from the oracle away from every FOV tie' % name
        # At and AtA: the flipped grid point's footprint in the output volume
        for op, arg in (('At', v), ('AtA', p)):
            ref = O.proj_apply(op, arg[None, None], po_o, method='denoising')[0, 0]
            out = apply(op, arg)
            bad = (out - ref).abs() > 1e-5 * float(ref.abs().max())
            assert not bool((bad & ~cover).any()), '%s (%s) differs from the oracle out of reach of every FOV tie' % (op, name)
    # and the ties are few: the operators agree on (nearly) the whole volume
    assert float(cover.float().mean()) < 0.5


@pytest.mark.parametrize('iperm', [9, 22, 41])
@pytest.mark.parametrize('trans,rot', GEOMS[:2] + [((1.0, -0.5, 0.05), (0.0, 0.0, 0.0))])
def test_reoriented_storage_disagreements_lie_within_reach_of_fov_ties(dev, iperm, trans, rot):
    """The same for observations STORED in another voxel order (sagittal / coronal / reflected: a signed axis
    permutation in the affine).  The plan relabels their axes and evaluates the grid coordinates from permuted,
    sign-flipped terms with the flip offset folded into the translation (api.hip: canonicalise) - equal to the
    reference's `lin @ ijk + off` to the last ulp or one off: a mask may flip where a coordinate
    ties with a threshold, nowhere else.  Thresholds hit by integer, half-voxel and 0.05-voxel translations."""
    import unires_amd as U
    dim_y = (26, 24, 22)
    mat_y = torch.eye(4, dtype=torch.float64)
    perm, flip = SIGNED_PERMS[iperm]
    dim_x, mat_x = orient_axes(dim_y, mat_y, perm, flip)
    rigid = rigid_matrix(list(trans), list(rot))
    po_o = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid)
    po_g = U._proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, device=dev)
    mat, dim_g = O.proj_matrix(po_o, 'denoising')
    near, g = _tie_points(mat, dim_g, dim_y, eps=1e-4)
    torch.manual_seed(7)
    p = torch.rand(dim_y) + 0.5
    v = torch.rand(dim_g) + 0.5
    cover = _reach(near, g, dim_y, reach=2)
    plan = ChannelPlan(dim_y, (1.0, 1.0, 1.0), [(po_g, 1.0)], 'denoising', True, device=dev)
    assert plan.repeat_info(0)['perm'] != (0, 1, 2) or any(plan.repeat_info(0)['flip'])
    ref = O.proj_apply('A', p[None, None], po_o, method='denoising')[0, 0]
    out = plan.proj_apply(0, 'A', p.to(dev)).cpu()
    bad = (out - ref).abs() > 1e-5 * float(ref.abs().max())
    assert not bool((bad & ~near).any()), 'A differs from the oracle away from every FOV tie'
    for op, arg in (('At', v), ('AtA', p)):
        ref = O.proj_apply(op, arg[None, None], po_o, method='denoising')[0, 0]
        out = plan.proj_apply(0, op, arg.to(dev)).cpu()
        bad = (out - ref).abs() > 1e-5 * float(ref.abs().max())
        assert not bool((bad & ~cover).any()), '%s differs from the oracle out of reach of every FOV tie' % op


@pytest.mark.parametrize('trans,rot', GEOMS[:3])
def test_super_resolution_disagreements_lie_within_reach_of_fov_ties(dev, trans, rot):
    """Same for A = conv_down . pull (thick slices along z): an x-space voxel may differ only if a tie
    point lies inside its slice-profile window."""
    import unires_amd as U
    dim_y, thick = (22, 20, 24), 3
    mat_y = torch.eye(4, dtype=torch.float64)
    mat_x = mat_y @ torch.diag(torch.tensor([1.0, 1.0, float(thick), 1.0], dtype=torch.float64))
    dim_x = (dim_y[0], dim_y[1], dim_y[2] // thick)
    rigid = rigid_matrix(list(trans), list(rot))
    po_o = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid)
    po_g = U._proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, device=dev)
    mat, dim_g = O.proj_matrix(po_o, 'super-resolution')
    near, _ = _tie_points(mat, dim_g, dim_y, eps=1e-4)
    torch.manual_seed(6)
    p = torch.rand(dim_y) + 0.5
    ref = O.proj_apply('A', p[None, None], po_o, method='super-resolution')[0, 0]
    # x-space voxel k reads grid points k * ratio .. k * ratio + K - 1 along z
    K = int(po_o.smo_ker.shape[-1])
    win = torch.zeros(dim_x, dtype=torch.bool)
    for k in range(dim_x[2]):
        win[:, :, k] = near[:, :, k * thick:k * thick + K].any(dim=2)
    plan = ChannelPlan(dim_y, (1.0, 1.0, 1.0), [(po_g, 1.0)], 'super-resolution', True, device=dev)
    for name, out in (('plan', plan.proj_apply(0, 'A', p.to(dev)).cpu()),
                      ('op-level', U._proj_apply('A', p[None, None].to(dev), po_g, method='super-resolution')[0, 0].cpu())):
        bad = (out - ref).abs() > 1e-5 * float(ref.abs().max())
        assert not bool((bad & ~win).any()), 'A (%s) differs from the oracle where no FOV tie is in the window' % name
