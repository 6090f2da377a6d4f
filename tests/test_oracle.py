"""Pins the CPU oracle (it cannot be pinned against nitorch itself - absent, see
oracle/__init__.py) with torch-native partial oracles, algebraic properties and
the few known answers the reference's notebooks print."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import make_problem, oracle_structs, rigid_matrix

torch.set_num_threads(4)


def test_get_gain_matches_notebook_trace():
    # demos/demo_single_channel.ipynb:173-175 prints gains inf, 1.0, 0.3567 for the
    # objective trace 5.481e6, 4.983e6, 4.706e6
    obj = torch.tensor([5.481e6, 4.983e6, 4.706e6], dtype=torch.float64)
    assert math.isinf(N.get_gain(obj[:1], 'decreasing'))
    assert abs(N.get_gain(obj[:2], 'decreasing') - 1.0) < 1e-12
    assert abs(N.get_gain(obj, 'decreasing') - 0.3567) < 2e-3


def test_proj_info_dims_match_notebooks():
    eye = torch.eye(4, dtype=torch.float64)
    # demos/demo_single_channel.ipynb:81-82: dim_y (181,217,181) -> dim_x (181,217,45) at 4 mm z
    D = torch.diag(torch.tensor([1, 1, 4, 1.], dtype=torch.float64))
    po = O.proj_info((181, 217, 181), eye, (181, 217, 45), eye @ D)
    assert po.ratio == (1, 1, 4) and po.dim_thick == 2
    assert po.dim_yx == (181, 217, 183)
    assert torch.allclose(po.smo_ker.flatten(),
                          torch.tensor([0, .125, .25, .25, .25, .125, 0]), atol=1e-7)
    # demos/demo_multi_channel.ipynb:109-113: thick along x and along y
    Dx = torch.diag(torch.tensor([4, 1, 1, 1.], dtype=torch.float64))
    po = O.proj_info((181, 217, 181), eye, (45, 217, 181), eye @ Dx)
    assert po.ratio == (4, 1, 1) and po.dim_thick == 0 and po.dim_yx == (183, 217, 181)


def test_rect_profile_known_values():
    assert N.smooth1d(0, 4.0) == pytest.approx([0, .125, .25, .25, .25, .125, 0], abs=1e-12)
    assert N.smooth1d(0, 6.0) == pytest.approx([0, 1 / 12, 1 / 6, 1 / 6, 1 / 6, 1 / 6, 1 / 6, 1 / 12, 0],
                                               abs=1e-12)
    for kind in (0, 1, 2):
        k = N.smooth1d(kind, 3.0)
        assert abs(sum(k) - 1) < 1e-12 and k == pytest.approx(k[::-1], abs=1e-12)


def test_pull_matches_grid_sample():
    torch.manual_seed(1)
    dim = (9, 11, 13)
    y = torch.rand((1, 1) + dim)
    M = rigid_matrix([0.4, -1.3, 2.2], [0.2, -0.1, 0.15])
    g = N.affine_grid(M.float(), (10, 9, 12))[None]
    a = N.grid_pull(y, g)
    gn = torch.stack([2 * g[..., 2] / (dim[2] - 1) - 1, 2 * g[..., 1] / (dim[1] - 1) - 1,
                      2 * g[..., 0] / (dim[0] - 1) - 1], -1)
    b = F.grid_sample(y, gn, mode='bilinear', padding_mode='zeros', align_corners=True)
    b = b * N._fov_mask(g[0], dim, N.FOV_TOL)
    assert (a - b).abs().max() < 5e-6


def test_push_is_autograd_adjoint_of_pull():
    torch.manual_seed(2)
    dim = (8, 7, 9)
    M = rigid_matrix([0.3, 0.8, -0.6], [0.1, 0.05, -0.2])
    g = N.affine_grid(M, (7, 8, 6))[None]
    y = torch.rand((1, 1) + dim, dtype=torch.float64, requires_grad=True)
    x = torch.rand((1, 1, 7, 8, 6), dtype=torch.float64)
    (N.grid_pull(y, g) * x).sum().backward()
    assert (y.grad - N.grid_push(x, g, dim)).abs().max() < 1e-12


def test_gradient_divergence_adjoint_and_stencil_rows():
    torch.manual_seed(3)
    dim = (6, 5, 7)
    vx = torch.tensor([1.0, 0.8, 1.5], dtype=torch.float64)
    y = torch.rand(dim, dtype=torch.float64)
    g = torch.rand((3,) + dim, dtype=torch.float64)
    lhs = (N.im_gradient(y, vx) * g).sum()
    rhs = (y * N.im_divergence(g, vx)).sum()
    assert abs(lhs - rhs) < 1e-12
    # 1-D rows of DtD: [1,-1] at 0, [-1,2,-1] interior, [-1,2] at n-1
    n = 5
    e = torch.eye(n, dtype=torch.float64)
    rows = torch.stack([O.DtD(e[i].reshape(n, 1, 1), torch.ones(3, dtype=torch.float64)).flatten()
                        for i in range(n)])
    # the singleton y and z axes (n=1) each contribute +1 on the diagonal
    rows = rows - 2 * torch.eye(n, dtype=torch.float64)
    expect = torch.tensor([[1., -1, 0, 0, 0], [-1, 2, -1, 0, 0], [0, -1, 2, -1, 0],
                           [0, 0, -1, 2, -1], [0, 0, 0, -1, 2]], dtype=torch.float64)
    assert torch.allclose(rows, expect)


@pytest.mark.parametrize('regime', ['sr', 'dn'])
def test_check_adjoint_float64(regime):
    prob = make_problem(regime=regime, scl=0.1 if regime == 'sr' else 0.0)
    x, _ = oracle_structs(prob)
    assert abs(O.check_adjoint(x[0][0].po, prob['method'])) < 1e-9


@pytest.mark.parametrize('regime', ['sr', 'dn', 'id'])
def test_matvec_is_symmetric_positive_definite(regime):
    prob = make_problem(dim_y=(6, 5, 4), regime=regime, thick=2, scl=0.05 if regime == 'sr' else 0)
    x, y = oracle_structs(prob)
    for xn in x[0]:
        xn.dat = xn.dat.double()
    n = 6 * 5 * 4
    vx = N.voxel_size(y[0].mat)
    cols = []
    for i in range(n):
        e = torch.zeros(n, dtype=torch.float64)
        e[i] = 1
        cols.append(O.proj('AtA', e.reshape(6, 5, 4), x[0], y[0], method=prob['method'],
                           do=prob['do_proj'], rho=prob['rho'], vx_y=vx).flatten())
    A = torch.stack(cols, 1)
    assert (A - A.T).abs().max() < 1e-12 * A.abs().max()
    assert torch.linalg.eigvalsh(0.5 * (A + A.T)).min() > 0
    # CG (tolerance 0 -> fixed iterations) converges to the dense solve
    b = torch.rand(n, dtype=torch.float64)
    sol = torch.linalg.solve(A, b)
    xcg = N.cg(lambda v: (A @ v.flatten()).reshape(v.shape), b.clone(), torch.zeros(n, dtype=torch.float64),
               max_iter=n, tolerance=0)
    assert (xcg - sol).norm() / sol.norm() < 1e-8


def test_cg_stop_branch_counts_extra_matvec():
    calls = {'n': 0}
    A = torch.diag(torch.linspace(1, 5, 8, dtype=torch.float64))

    def op(v):
        calls['n'] += 1
        return A @ v
    b = torch.ones(8, dtype=torch.float64)
    _, n_it, obj = N.cg(op, b, torch.zeros(8, dtype=torch.float64), max_iter=3, tolerance=1e-30,
                        stop='max_gain', return_info=True)
    assert n_it == 3 and calls['n'] == 1 + 1 + 2 * 3 and len(obj) == 4
    assert all(obj[i + 1] <= obj[i] for i in range(3))  # energy decreases monotonically


def test_update_y_reduces_objective():
    prob = make_problem(dim_y=(12, 10, 9), n_channels=2, thick=3, scl=0.05)
    x, y = oracle_structs(prob)
    y, info = O.update_y(x, y, prob['z'].clone(), prob['w'].clone(), torch.tensor(prob['rho']),
                         prob['method'], prob['do_proj'], return_info=True)
    for n_it, obj in info:
        assert 1 <= n_it <= 20 and obj[-1] < obj[0]


@pytest.mark.parametrize('fwhm', [1.5, 2.0, 3.0, 6.0])
def test_slice_profiles_are_the_profile_convolved_with_the_linear_basis(fwhm):
    """Independent of the restatement's own quadrature and of the closed form it uses for the
    Gaussian (SPM's spm_smoothkern erf / exp expression): every tap of smooth1d equals the integral
    of profile(t) * tri(x - t) computed by scipy.integrate.quad, after the same normalisation."""
    import math
    from scipy.integrate import quad
    tri = lambda t: max(0.0, 1.0 - abs(t))
    s2 = (fwhm / math.sqrt(8.0 * math.log(2.0))) ** 2
    profiles = {0: (lambda t: (1.0 / fwhm) if abs(t) <= fwhm / 2 else 0.0, fwhm / 2),
                1: (lambda t: tri(t / fwhm) / fwhm, fwhm),
                2: (lambda t: math.exp(-0.5 * t * t / s2) / math.sqrt(2.0 * math.pi * s2), 12.0 * math.sqrt(s2))}
    for kind, (prof, reach) in profiles.items():
        ker = N.smooth1d(kind, fwhm)
        L = (len(ker) - 1) // 2
        ref = []
        for x in range(-L, L + 1):
            lo, hi = max(-reach, x - 1.0), min(reach, x + 1.0)
            pts = sorted({p for p in (-fwhm / 2, fwhm / 2, 0.0, float(x), -reach, reach) if lo < p < hi})
            ref.append(quad(lambda t: prof(t) * tri(x - t), lo, hi, points=pts or None, epsabs=1e-13, epsrel=1e-12)[0]
                       if hi > lo else 0.0)
        tot = sum(ref)
        for a, b in zip(ker, ref):
            assert abs(a - b / tot) < 2e-9, (kind, fwhm)


def test_cg_iterates_are_those_of_scipy_cg_and_converge_to_the_solve():
    """The restated cg against code that shares nothing with it: after k iterations its x equals the
    k-th iterate of scipy.sparse.linalg.cg (same Krylov recurrences, float64) and after enough of them
    numpy's direct solve; the preconditioned form agrees with scipy's M argument."""
    import numpy as np
    from scipy.sparse.linalg import cg as scipy_cg
    rng = np.random.default_rng(3)
    n = 40
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T + n * np.eye(n)
    b = rng.standard_normal(n)
    d = np.diag(A).copy()
    At, bt = torch.tensor(A), torch.tensor(b)
    for use_pre in (False, True):
        for k in (1, 3, 7):
            its = []
            scipy_cg(A, b, x0=np.zeros(n), maxiter=k, rtol=0.0, atol=0.0,
                     M=np.diag(1.0 / d) if use_pre else None, callback=lambda xk: its.append(xk.copy()))
            x = N.cg(lambda v: At @ v, bt, torch.zeros(n, dtype=torch.float64), max_iter=k, tolerance=0,
                     precond=(lambda r: r / torch.tensor(d)) if use_pre else (lambda r: r))
            assert len(its) == k
            assert np.abs(x.numpy() - its[-1]).max() < 1e-12
        x = N.cg(lambda v: At @ v, bt, torch.zeros(n, dtype=torch.float64), max_iter=80, tolerance=0,
                 precond=(lambda r: r / torch.tensor(d)) if use_pre else (lambda r: r))
        assert np.abs(x.numpy() - np.linalg.solve(A, b)).max() < 1e-10


def test_grid_grad_is_the_derivative_of_grid_sample():
    """The restated grid_grad (what the rigid Gauss-Newton differentiates through,
    unires/_update.py:508) against autograd through torch's own trilinear sampler: d pull / d g in voxel
    coordinates = d grid_sample / d g_normalised * 2 / (n - 1), away from the volume's faces where
    padding and the FOV mask enter."""
    torch.manual_seed(4)
    dim = (9, 11, 13)
    y = torch.rand((1, 1) + dim, dtype=torch.float64)
    M = rigid_matrix([0.4, -0.3, 0.6], [0.05, -0.04, 0.03])
    g = N.affine_grid(M, (6, 7, 8))[None].clone()
    g = g.clamp(min=0.3)  # keep every sample strictly inside
    for d, n in enumerate(dim):
        g[..., d] = g[..., d].clamp(max=n - 1.3)
    g = g + 1e-3  # off the integer lattice: the derivative of a trilinear sample jumps there
    g.requires_grad_(True)
    gn = torch.stack([2 * g[..., 2] / (dim[2] - 1) - 1, 2 * g[..., 1] / (dim[1] - 1) - 1,
                      2 * g[..., 0] / (dim[0] - 1) - 1], -1)
    F.grid_sample(y, gn, mode='bilinear', padding_mode='zeros', align_corners=True).sum().backward()
    ours = N.grid_grad(y, g.detach())[0, 0]
    assert (ours - g.grad[0]).abs().max() < 1e-10
