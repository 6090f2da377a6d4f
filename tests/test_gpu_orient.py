"""Orientation-general operators.  The reference passes whatever affine a file
carries straight into the operator (unires/_core.py:145-168 resets CT affines only, _util.py:134-197
keeps ``mat`` as read, _project.py:147-159 / 217-232 build a dense grid from it), so observations
STORED sagittally or coronally (voxel axes permuted against the world axes) or reflected (LAS vs RAS,
det < 0) reach A as a signed axis permutation composed with the rigid.  Every one of the 48 signed
permutations, for the pull-only and the thick-slice regimes, against the oracle - which, like the
reference, knows nothing about orientation: it samples a dense grid."""
import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import (SIGNED_PERMS, gpu_structs, make_problem, oracle_structs, rel_err,
                           run_gpu_update_y, run_oracle_update_y)

pytestmark = pytest.mark.gpu

GATE = 1e-4


def _problem(i, regime, seed, n_repeats=1, **kw):
    """Signed permutation i for repeat 0 (a different one for a second repeat); the thick axis - before
    storage order is applied - cycles with i, so that after it every voxel axis carries thick slices."""
    orient = [SIGNED_PERMS[i], SIGNED_PERMS[(7 * i + 5) % 48]]
    if regime == 'sr':
        return make_problem(dim_y=(16, 14, 12), n_channels=1, thick=3, regime='sr', thick_axes=[i % 3],
                            n_repeats=n_repeats, orient=orient, seed=seed, **kw)
    return make_problem(dim_y=(13, 11, 12), n_channels=1, regime='dn', rot=0.08, trans=1.2,
                        n_repeats=n_repeats, orient=orient, seed=seed, **kw)


@pytest.mark.parametrize('regime', ['sr', 'dn'])
@pytest.mark.parametrize('i', range(48))
def test_operators_under_every_signed_permutation(dev, regime, i):
    """A, At, AtA of one repeat: op-level kernels (_proj_apply) and the plan's (_proj), the reference's
    adjoint harness, and which kernels the plan runs the operator on."""
    import unires_amd as U
    from unires_amd._project import _channel_plan
    prob = _problem(i, regime, seed=100 + i, scl=0.1 if regime == 'sr' else 0.0)
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    po_o, po_g = xo[0][0].po, xg[0][0].po
    assert tuple(po_g.dim_x) == tuple(po_o.dim_x) and tuple(po_g.dim_yx or ()) == tuple(po_o.dim_yx or ())
    assert int(po_g.dim_thick) == int(po_o.dim_thick) and tuple(po_g.ratio) == tuple(po_o.ratio)
    torch.manual_seed(i)
    yv = torch.rand((1, 1) + prob['dim_y'])
    xv = torch.rand((1, 1) + tuple(po_o.dim_x))
    for operator, v in (('A', yv), ('At', xv), ('AtA', yv)):
        ref = O.proj_apply(operator, v, po_o, method=prob['method'])
        out = U._proj_apply(operator, v.to(dev), po_g, method=prob['method']).cpu()
        assert out.shape == ref.shape
        assert rel_err(out, ref) < 2e-5, ('op-level', operator)
    val = U._check_adjoint(po_g, prob['method'])
    assert abs(val) < 1e-5 * float(torch.tensor(po_o.dim_x).prod())
    for operator, v in (('A', yv), ('At', xv)):
        ref = O.proj(operator, v[0, 0], xo[0], yo[0], method=prob['method'], n=0)
        out = U._proj(operator, v[0, 0].to(dev), xg[0], yg[0], method=prob['method'], n=0).cpu()
        assert out.shape == ref.shape
        assert rel_err(out, ref) < 2e-5, ('plan', operator)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    p = yv[0, 0] * 100
    ref = O.proj('AtA', p, xo[0], yo[0], method=prob['method'], rho=rho, vx_y=vx)
    out = U._proj('AtA', p.to(dev), xg[0], yg[0], method=prob['method'], rho=rho, vx_y=vx).cpu()
    assert rel_err(out, ref) < 2e-5, 'plan AtA'
    # no orientation is left to the general-geometry fallbacks: the LDS-window pull and the
    # schedule-driven splat serve every one of them
    info = _channel_plan(xg[0], yg[0], prob['method'], True, vx).repeat_info(0)
    assert info['pull2'] and info['splat2_axis'] is not None, info
    perm, flip = SIGNED_PERMS[i]
    if i == 0:
        assert info['perm'] == (0, 1, 2) and info['flip'] == (0, 0, 0)
    else:
        assert info['perm'] != (0, 1, 2) or any(info['flip'])


@pytest.mark.parametrize('regime', ['sr', 'dn'])
@pytest.mark.parametrize('tol', [0.0, 1e-3])
@pytest.mark.parametrize('i', range(0, 48, 1))
def test_update_y_under_every_signed_permutation(dev, regime, tol, i):
    """The grading gate with two repeats of DIFFERENT orientation in the channel, even / odd slice
    scaling on: same start, same inputs, same CG settings -> 1e-4 relative, equal iteration counts."""
    if tol and i % 4:
        pytest.skip('the reference-default stopping rule on every fourth orientation')
    prob = _problem(i, regime, seed=300 + i, n_repeats=2, scl=0.07 if regime == 'sr' else 0.0)
    y_ref, info_ref = run_oracle_update_y(prob, max_iter=12, tol=tol)
    y_gpu, info_gpu = run_gpu_update_y(prob, dev, max_iter=12, tol=tol)
    assert info_gpu[0][0] == info_ref[0][0], 'realised CG iterations differ'
    assert rel_err(y_gpu[0].cpu(), y_ref[0]) < GATE
    if tol:
        o_gpu = torch.tensor(info_gpu[0][1], dtype=torch.float64)
        assert torch.allclose(o_gpu, info_ref[0][1], rtol=1e-5, atol=0)


def test_reoriented_aligned_observation_takes_the_one_kernel_matvec(dev):
    """Sagittal-stored, reflected, NOT rotated: after relabelling the observation is grid-aligned, so the
    streaming one-kernel matvec (aligned.hip) serves it - same numbers as the oracle's dense grid."""
    import unires_amd as U
    prob = make_problem(dim_y=(16, 14, 24), n_channels=2, thick=3, regime='sr', thick_axes=[2, 2], rot=0.0,
                        trans=0.0, scl=0.1, orient=[((2, 0, 1), (1, 0, 1)), ((1, 2, 0), (0, 1, 0))], seed=7)
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    torch.manual_seed(2)
    for c in range(2):
        p = torch.rand(prob['dim_y']) * 100
        ref = O.proj('AtA', p, xo[c], yo[c], method=prob['method'], rho=rho, vx_y=vx)
        out = U._proj('AtA', p.to(dev), xg[c], yg[c], method=prob['method'], rho=rho, vx_y=vx).cpu()
        assert rel_err(out, ref) < 2e-5
    y_ref, info_ref = run_oracle_update_y(prob, max_iter=10, tol=0.0)
    y_gpu, info_gpu = run_gpu_update_y(prob, dev, max_iter=10, tol=0.0)
    for c in range(2):
        assert rel_err(y_gpu[c].cpu(), y_ref[c]) < GATE


@pytest.mark.parametrize('i', [9, 22, 35, 47])
def test_isotropic_downsampling_and_default_profiles_reoriented(dev, i):
    """BASELINE config 4's shape (every axis coarser, Gaussian in-plane profile = the reference's
    default, struct.py:95-96) stored in a permuted / reflected order: hybrid and separable passes."""
    prob = make_problem(dim_y=(16, 14, 12), n_channels=1, thick=2, regime='sr', iso=True, prof_ip=2, prof_tp=0,
                        vx_y=0.5, scl=0.1, orient=[SIGNED_PERMS[i]], seed=500 + i)
    y_ref, info_ref = run_oracle_update_y(prob, max_iter=10, tol=0.0)
    y_gpu, info_gpu = run_gpu_update_y(prob, dev, max_iter=10, tol=0.0)
    assert rel_err(y_gpu[0].cpu(), y_ref[0]) < GATE
