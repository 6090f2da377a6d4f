"""GPU parity of the fused path (plan / matvec / RHS / CG / y-update) against
the CPU oracle on the same seeded inputs; float32 gate 1e-4 relative
(BASELINE.json north_star)."""
import os

import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import (gpu_structs, make_problem, oracle_structs, rel_err, run_gpu_update_y,
                           run_oracle_update_y)

pytestmark = pytest.mark.gpu

GATE = 1e-4  # north_star: float32 agreement with the reference CPU path, relative

CASES = {
    'sr_z_1ch': dict(dim_y=(16, 14, 12), n_channels=1, thick=3, regime='sr', thick_axes=[2]),
    'sr_3ch_axes': dict(dim_y=(20, 18, 16), n_channels=3, thick=4, regime='sr', scl=0.1),
    'sr_2rep': dict(dim_y=(14, 12, 15), n_channels=2, thick=2, regime='sr', n_repeats=2, scl=0.05),
    'sr_gauss_tri': dict(dim_y=(12, 12, 12), n_channels=1, thick=2, regime='sr', prof_tp=1, prof_ip=2),
    'sr_aniso': dict(dim_y=(14, 10, 12), n_channels=1, thick=3, regime='sr', aniso=(0.8, 1.0, 1.3)),
    'sr_iso2_gauss': dict(dim_y=(16, 14, 12), n_channels=2, thick=2, regime='sr', iso=True, prof_ip=2,
                          prof_tp=0, vx_y=0.5),
    'sr_iso3_rect': dict(dim_y=(15, 15, 12), n_channels=1, thick=3, regime='sr', iso=True, rot=0.02),
    # profiles along all three axes with even / odd slice scaling: along x (rides with the 1-D x / y
    # passes of the hybrid path) and along z (rides with the fused z profile)
    'sr_iso2_scl_x': dict(dim_y=(16, 14, 12), n_channels=1, thick=2, regime='sr', iso=True, scl=0.1),
    'sr_223_scl_z': dict(dim_y=(14, 12, 18), n_channels=2, thick=3, regime='sr', iso=(2, 2, 3), scl=0.1,
                         n_repeats=2),
    # x-space z extent a multiple of 4: the 16-byte 1-D passes, and the fused x-y pass of the hybrid path
    'sr_iso2_z8': dict(dim_y=(18, 14, 16), n_channels=2, thick=2, regime='sr', iso=True, scl=0.1),
    'sr_iso3_z8': dict(dim_y=(15, 18, 24), n_channels=1, thick=3, regime='sr', iso=True, rot=0.02),
    # 6 mm slices along z, 11 and 21 of them: the window pull takes 11 conv windows per chunk instead of
    # the 10 its 64 lanes hold (one chunk instead of two, two instead of three) and samples the three grid
    # points per row beyond the lanes in a pass of their own
    'sr_thick6_z66': dict(dim_y=(14, 12, 66), n_channels=2, thick=6, regime='sr', thick_axes=[2, 2], scl=0.1),
    'sr_thick6_z126': dict(dim_y=(10, 9, 126), n_channels=1, thick=6, regime='sr', thick_axes=[2], rot=0.03),
    'sr_aligned': dict(dim_y=(16, 14, 24), n_channels=2, thick=3, regime='sr', thick_axes=[2, 2], rot=0.0,
                       trans=0.0, scl=0.1),
    # translated by a fraction of a voxel, not rotated: the factorised one-kernel matvec (shift.hip); nz % 4 == 0
    'sr_shift': dict(dim_y=(16, 14, 24), n_channels=2, thick=3, regime='sr', thick_axes=[2, 2], rot=0.0,
                     trans=1.7, scl=0.1),
    'sr_shift_big': dict(dim_y=(13, 11, 40), n_channels=1, thick=4, regime='sr', thick_axes=[2], rot=0.0,
                         trans=6.3),
    # 256-voxel lines: the fast form of the x-marching kernel (every lane inside the line, lane-window z operator),
    # two lines per wave / one (odd ny); the same kernel serves integer shifts (rigid = I) there
    'sr_shift_z256': dict(dim_y=(12, 10, 256), n_channels=2, thick=6, regime='sr', thick_axes=[2, 2], rot=0.0,
                          trans=2.3, scl=0.1),
    'sr_shift_z256_odd': dict(dim_y=(9, 11, 256), n_channels=1, thick=4, regime='sr', thick_axes=[2], rot=0.0, trans=1.1),
    'sr_aligned_z256': dict(dim_y=(10, 12, 256), n_channels=1, thick=6, regime='sr', thick_axes=[2], rot=0.0, trans=0.0,
                            scl=0.05),
    'dn_shift': dict(dim_y=(15, 13, 12), n_channels=2, regime='dn', rot=0.0, trans=2.4),
    'dn_2ch': dict(dim_y=(15, 13, 11), n_channels=2, regime='dn', rot=0.1, trans=1.5),
    'dn_2rep': dict(dim_y=(12, 12, 10), n_channels=1, regime='dn', n_repeats=2),
    'id_1ch': dict(dim_y=(18, 17, 13), n_channels=1, regime='id'),
    'id_2rep': dict(dim_y=(9, 8, 70), n_channels=2, regime='id', n_repeats=2),
    # observations STORED sagittally / coronally / reflected (voxel axes a signed permutation of the world's:
    # what a NIfTI file hands the reference as it is, _util.py:134-197); tests/test_gpu_orient.py sweeps all 48
    'sr_orient': dict(dim_y=(18, 16, 14), n_channels=2, thick=3, regime='sr', scl=0.05, n_repeats=2,
                      orient=[((2, 0, 1), (1, 0, 0)), ((1, 2, 0), (0, 1, 1)), ((0, 1, 2), (1, 0, 0)),
                              ((2, 1, 0), (0, 0, 1))]),
    'dn_orient': dict(dim_y=(15, 13, 11), n_channels=2, regime='dn', rot=0.1, trans=1.5,
                      orient=[((1, 0, 2), (0, 0, 0)), ((2, 0, 1), (1, 1, 1))]),
    # many-tap profiles at ratio 2 with x-space z a multiple of 4: the separable marching passes and the one-pass
    # conv_down_x / conv_up_x of A^T A with conv_down_y in front (k_conv_ydown_xdownup2, k_conv1d_downup2_m) and the
    # y + z conv_up (k_conv_up_yz2) with 3, 5 and 11 taps on every axis (the channels' thick axes differ;
    # tools/which_kernels.py lists a case's kernels)
    'sr_iso2_gauss_v4': dict(dim_y=(20, 24, 32), n_channels=3, thick=2, regime='sr', iso=True, prof_ip=2, scl=0.05),
    'sr_iso2_tri_gauss_v4': dict(dim_y=(20, 24, 32), n_channels=3, thick=2, regime='sr', iso=True, prof_ip=2, prof_tp=1),
    'sr_iso2_allgauss_v4': dict(dim_y=(24, 20, 32), n_channels=2, thick=2, regime='sr', iso=True, prof_ip=2, prof_tp=2,
                                scl=0.05),
    # ... and the observation's thick axis along y and z after the plan's relabelling (3 y taps with 11 x taps;
    # z-profile splat with the 11 x 11 in-plane part through the fused x / y kernel)
    'sr_iso2_gauss_orient_v4': dict(dim_y=(24, 24, 32), n_channels=3, thick=2, regime='sr', iso=True, prof_ip=2, scl=0.05,
                                    orient=[((0, 1, 2), (0, 0, 0)), ((1, 0, 2), (0, 1, 0)), ((2, 1, 0), (0, 0, 1))]),
    # ratio 2 with the rect profile (BASELINE config 4's operator): conv_down_y / x and conv_up_x / y of A^T A in one
    # kernel (k_conv_ydown_xdownup2<3, 3, 2, 2>), the z part in the pull and the splat
    'sr_iso2_rect_v4': dict(dim_y=(24, 20, 32), n_channels=2, thick=2, regime='sr', iso=True, scl=0.1),
    'sr_iso2_tri_v4': dict(dim_y=(24, 20, 32), n_channels=1, thick=2, regime='sr', iso=True, prof_ip=1, prof_tp=1),
}


@pytest.mark.parametrize('case', list(CASES))
def test_proj_apply_and_adjoint(dev, case):
    import unires_amd as U
    prob = make_problem(seed=11, **CASES[case])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    torch.manual_seed(5)
    for c in range(len(xo)):
        for n in range(len(xo[c])):
            po_o, po_g = xo[c][n].po, xg[c][n].po
            op = 'none' if not prob['do_proj'] else None
            yv = torch.rand((1, 1) + prob['dim_y'])
            xv = torch.rand((1, 1) + tuple(po_o.dim_x))
            for operator, v in (('A', yv), ('At', xv), ('AtA', yv)):
                ref = O.proj_apply(op or operator, v, po_o, method=prob['method'])
                out = U._proj_apply(op or operator, v.to(dev), po_g, method=prob['method']).cpu()
                assert out.shape == ref.shape
                assert rel_err(out, ref) < 2e-5, (operator, c, n)
            if prob['do_proj']:  # the reference's own harness, unires/_project.py:27-51
                val = U._check_adjoint(po_g, prob['method'])
                scale = float(torch.tensor(po_o.dim_x).prod())
                assert abs(val) < 1e-5 * scale
            # fused per-repeat operators of the plan agree with the composed ones
            for operator, v in (('A', yv), ('At', xv)):
                ref = O.proj(operator, v[0, 0], xo[c], yo[c], method=prob['method'], do=prob['do_proj'], n=n)
                out = U._proj(operator, v[0, 0].to(dev), xg[c], yg[c], method=prob['method'],
                              do=prob['do_proj'], n=n).cpu()
                assert rel_err(out, ref) < 2e-5, ('plan', operator, c, n)


@pytest.mark.parametrize('case', list(CASES))
def test_matvec_and_rhs(dev, case):
    import unires_amd as U
    from unires_amd._project import _channel_plan
    prob = make_problem(seed=12, **CASES[case])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    torch.manual_seed(6)
    for c in range(len(xo)):
        p = torch.rand(prob['dim_y']) * 100
        ref = O.proj('AtA', p, xo[c], yo[c], method=prob['method'], do=prob['do_proj'], rho=rho, vx_y=vx)
        out = U._proj('AtA', p.to(dev), xg[c], yg[c], method=prob['method'], do=prob['do_proj'],
                      rho=rho, vx_y=vx).cpu()
        assert rel_err(out, ref) < 2e-5
        # fused dot product sum(p * q) in float64
        plan = _channel_plan(xg[c], yg[c], prob['method'], prob['do_proj'], vx)
        dot = torch.zeros((), dtype=torch.float64, device=dev)
        q = plan.matvec(p.to(dev), float(rho), float(yg[c].lam), dot=dot)
        ref_dot = torch.sum(p * ref, dtype=torch.float64)
        assert abs(dot.item() - ref_dot.item()) < 2e-5 * abs(ref_dot.item())
        # RHS b = sum tau At x - lam Dt(w - rho z)
        ref_b = O.y_rhs(xo[c], yo[c], prob['z'][c], prob['w'][c], rho, vx, prob['method'], prob['do_proj'])
        b = plan.rhs([xn.dat for xn in xg[c]], prob['w'][c].to(dev), prob['z'][c].to(dev),
                     float(rho), float(yg[c].lam)).cpu()
        assert rel_err(b, ref_b) < 2e-5
        # kept data term: same b, recomputed when an observation changes in place
        out = torch.empty(prob['dim_y'], device=dev)
        for _ in range(2):
            plan.rhs_cached([xn.dat for xn in xg[c]], prob['w'][c].to(dev), prob['z'][c].to(dev),
                            float(rho), float(yg[c].lam), out)
            assert rel_err(out.cpu(), ref_b) < 2e-5
        xg[c][0].dat.mul_(1.5)
        xo[c][0].dat = xo[c][0].dat * 1.5
        ref_b2 = O.y_rhs(xo[c], yo[c], prob['z'][c], prob['w'][c], rho, vx, prob['method'], prob['do_proj'])
        plan.rhs_cached([xn.dat for xn in xg[c]], prob['w'][c].to(dev), prob['z'][c].to(dev),
                        float(rho), float(yg[c].lam), out)
        assert rel_err(out.cpu(), ref_b2) < 2e-5


@pytest.mark.parametrize('case', list(CASES))
@pytest.mark.parametrize('tol', [0.0, 1e-3])
def test_update_y_matches_oracle(dev, case, tol):
    """The grading gate: same start, same inputs, same CG settings -> 1e-4 relative."""
    prob = make_problem(seed=13, **CASES[case])
    y_ref, info_ref = run_oracle_update_y(prob, max_iter=20, tol=tol)
    y_gpu, info_gpu = run_gpu_update_y(prob, dev, max_iter=20, tol=tol)
    for c in range(len(y_ref)):
        assert info_gpu[c][0] == info_ref[c][0], 'realised CG iterations differ'
        assert rel_err(y_gpu[c].cpu(), y_ref[c]) < GATE
        if tol:
            o_ref = info_ref[c][1]
            o_gpu = torch.tensor(info_gpu[c][1], dtype=torch.float64)
            assert torch.allclose(o_gpu, o_ref, rtol=1e-5, atol=0)


@pytest.mark.parametrize('case', ['sr_z_1ch', 'sr_3ch_axes', 'sr_aligned', 'dn_2ch', 'id_1ch'])
def test_jacobi_preconditioner_matches_oracle(dev, case):
    """_precond (unires/_update.py:80-102, commented out at :136): diagonal and PCG iterates."""
    import unires_amd as U
    prob = make_problem(seed=21, **CASES[case])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    if prob['do_proj']:  # the reference's _precond always applies AtA (no `do` switch)
        for c in range(len(xo)):
            pre_o = O.precond(xo[c], yo[c], rho, prob['method'])
            pre_g = U._precond(xg[c], yg[c], float(rho), sett)
            v = torch.rand(prob['dim_y']) + 0.5
            assert rel_err(pre_g(v.to(dev)).cpu(), pre_o(v)) < 2e-5
    for tol in (0.0, 1e-3):
        if not prob['do_proj']:
            break
        y_ref, info_ref = run_oracle_update_y(prob, max_iter=12, tol=tol, jacobi=True)
        y_gpu, info_gpu = run_gpu_update_y(prob, dev, max_iter=12, tol=tol, precond='jacobi')
        for c in range(len(y_ref)):
            assert info_gpu[c][0] == info_ref[c][0], 'realised PCG iterations differ'
            assert rel_err(y_gpu[c].cpu(), y_ref[c]) < GATE


def test_jacobi_identity_regime_and_errors(dev):
    import unires_amd as U
    from unires_amd._project import _channel_plan
    prob = make_problem(seed=22, **CASES['id_1ch'])
    xg, yg, sett = gpu_structs(prob, dev)
    plan = _channel_plan(xg[0], yg[0], sett.method, sett.do_proj)
    M = torch.empty(prob['dim_y'], device=dev)
    rho, lam = 0.7, float(yg[0].lam)
    plan.precond_build(rho, lam, out=M)
    vx = N.voxel_size(prob['mat_y']).float()
    want = float(xg[0][0].tau) + 2 * rho * lam ** 2 * float(vx.square().reciprocal().sum())
    assert torch.allclose(M.cpu(), torch.full(prob['dim_y'], want), rtol=1e-6)
    b = torch.rand(prob['dim_y'], device=dev)
    x = torch.zeros_like(b)
    with pytest.raises(ValueError, match="unires_precond_build"):  # built for another rho
        plan.cg(b, x, rho + 0.1, lam, precond='jacobi')
    with pytest.raises(ValueError):
        plan.cg(b, x, rho, lam, precond='multigrid')
    with pytest.raises(ValueError, match='unires_precond_build'):
        plan.cg(b, x, rho, lam, precond='fft')  # Jacobi was built, not FFT
    prob2 = make_problem(seed=23, **CASES['dn_2rep'])
    x2, y2, sett2 = gpu_structs(prob2, dev)
    with pytest.raises(ValueError, match='one repeat per contrast'):
        U._precond(x2[0], y2[0], 1.0, sett2)


@pytest.mark.parametrize('case', ['sr_3ch_axes', 'sr_2rep', 'dn_2ch', 'id_1ch', 'id_2rep'])
def test_fft_preconditioner(dev, case):
    """Build-side FFT-diagonal preconditioner (rocFFT): operator parity with the torch.fft
    restatement, symmetry / positivity, PCG iterates against the oracle's PCG, and the point of
    it - a smaller residual than plain CG after the same number of iterations."""
    import unires_amd as U
    from unires_amd._project import _channel_plan
    prob = make_problem(seed=61, **CASES[case])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    torch.manual_seed(3)
    for c in range(len(xo)):
        plan = _channel_plan(xg[c], yg[c], sett.method, sett.do_proj)
        plan.precond_build(float(rho), float(yg[c].lam), mode='fft')
        pre_o = O.fft_precond(xo[c], yo[c], rho, prob['method'], prob['do_proj'])
        u, v = torch.rand(prob['dim_y']), torch.rand(prob['dim_y'])
        Pu, Pv = plan.precond_apply(u.to(dev)).cpu(), plan.precond_apply(v.to(dev)).cpu()
        assert rel_err(Pu, pre_o(u)) < 2e-5
        s1, s2 = (Pu.double() * v.double()).sum(), (u.double() * Pv.double()).sum()
        assert abs(s1 - s2) < 1e-5 * abs(s1) and (Pu.double() * u.double()).sum() > 0
    y_ref, info_ref = run_oracle_update_y(prob, max_iter=8, tol=0.0, fft=True)
    y_pcg, _ = run_gpu_update_y(prob, dev, max_iter=8, tol=0.0, precond='fft')
    y_cg, _ = run_gpu_update_y(prob, dev, max_iter=8, tol=0.0, precond='none')
    zo, wo = prob['z'], prob['w']
    for c in range(len(y_ref)):
        assert rel_err(y_pcg[c].cpu(), y_ref[c]) < GATE
        vx = N.voxel_size(prob['mat_y']).float()
        b = O.y_rhs(xo[c], yo[c], zo[c], wo[c], rho, vx, prob['method'], prob['do_proj'])
        lhs = lambda d: O.proj('AtA', d, xo[c], yo[c], method=prob['method'], do=prob['do_proj'],
                               rho=rho, vx_y=vx)
        r_pcg = (b - lhs(y_pcg[c].cpu())).norm()
        r_cg = (b - lhs(y_cg[c].cpu())).norm()
        assert r_pcg < r_cg, (c, float(r_pcg), float(r_cg))


def test_recurred_objective_stops_at_the_same_iteration(dev):
    prob = make_problem(seed=14, **CASES['sr_3ch_axes'])
    _, info_a = run_gpu_update_y(prob, dev, tol=1e-3, stop='max_gain')
    y_b, info_b = run_gpu_update_y(prob, dev, tol=1e-3, stop='max_gain_recurred')
    y_ref, _ = run_oracle_update_y(prob, tol=1e-3)
    for c in range(3):
        assert info_a[c][0] == info_b[c][0]
        assert rel_err(y_b[c].cpu(), y_ref[c]) < GATE


def test_residual_stop_mode(dev):
    """nitorch's other objective branch (stop='e': sqrt(r.z))."""
    from unires_amd._project import _channel_plan
    prob = make_problem(seed=15, **CASES['dn_2ch'])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    b = O.y_rhs(xo[0], yo[0], prob['z'][0], prob['w'][0], rho, vx, prob['method'], True)
    lhs = lambda d: O.proj('AtA', d, xo[0], yo[0], method=prob['method'], rho=rho, vx_y=vx)
    xr, n_ref, obj_ref = N.cg(lhs, b, yo[0].dat.clone(), max_iter=20, tolerance=1e-2, stop='E',
                              return_info=True)
    plan = _channel_plan(xg[0], yg[0], prob['method'], True, vx)
    n_gpu, obj = plan.cg(b.to(dev), yg[0].dat, float(rho), float(yg[0].lam), 20, 1e-2, stop='e')
    assert n_gpu == n_ref and rel_err(yg[0].dat.cpu(), xr) < GATE
    assert torch.allclose(torch.tensor(obj, dtype=torch.float64), obj_ref, rtol=1e-4)


def test_plan_rebuilds_when_the_rigid_changes(dev):
    import unires_amd as U
    from tests.helpers import rigid_matrix
    prob = make_problem(seed=16, **CASES['sr_z_1ch'])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    vx = N.voxel_size(prob['mat_y']).float()
    p = torch.rand(prob['dim_y'])
    U._proj('AtA', p.to(dev), xg[0], yg[0], rho=1.0, vx_y=vx)
    new = rigid_matrix([0.2, 0.1, -0.3], [0.01, 0.04, -0.02])
    xo[0][0].po.rigid = new
    xg[0][0].po.rigid = new
    ref = O.proj('AtA', p, xo[0], yo[0], rho=torch.tensor(1.0), vx_y=vx)
    out = U._proj('AtA', p.to(dev), xg[0], yg[0], rho=1.0, vx_y=vx).cpu()
    assert rel_err(out, ref) < 2e-5


def test_large_volume_properties(dev):
    """Size-independent properties at a BASELINE-scale volume (256^3 would take the
    oracle minutes per operator; properties need no oracle):
    adjointness <Ay,x> = <y,Atx>, symmetry <Ap,q> = <p,Aq>, positivity, CG descent."""
    import unires_amd as U
    from unires_amd._project import _channel_plan
    from tests.helpers import rigid_matrix
    dim_y = (256, 256, 256)
    eye = torch.eye(4, dtype=torch.float64)
    D = torch.diag(torch.tensor([1, 1, 6, 1.], dtype=torch.float64))
    po = U._proj_info(dim_y, eye, (256, 256, 42), eye @ D,
                      rigid=rigid_matrix([2.0, -3.0, 1.0], [0.05, -0.08, 0.03]), device=dev)
    assert po.dim_yx == (256, 256, 255) and len(po.smo_ker_1d[2]) == 9
    val = U._check_adjoint(po, 'super-resolution')
    assert abs(val) < 1e-5 * 256 * 256 * 42
    g = torch.Generator().manual_seed(0)
    x = [U._input(torch.rand((256, 256, 42), generator=g).to(dev), eye @ D, 1.8e-4, po)]
    y = U._output(torch.zeros(dim_y, device=dev), eye, 0.006)
    plan = _channel_plan(x, y, 'super-resolution', True)
    p = torch.rand(dim_y, generator=g).to(dev)
    q = torch.rand(dim_y, generator=g).to(dev)
    Ap = plan.matvec(p, 0.9, 0.006)
    Aq = plan.matvec(q, 0.9, 0.006)
    s1 = torch.sum(Ap * q, dtype=torch.float64).item()
    s2 = torch.sum(p * Aq, dtype=torch.float64).item()
    assert abs(s1 - s2) < 1e-5 * abs(s1)
    assert torch.sum(Ap * p, dtype=torch.float64).item() > 0
    b = plan.rhs([x[0].dat], torch.zeros((3,) + dim_y, device=dev), torch.zeros((3,) + dim_y, device=dev),
                 0.9, 0.006)
    n_it, obj = plan.cg(b, y.dat, 0.9, 0.006, max_iter=20, tolerance=1e-3)
    assert 1 <= n_it <= 20 and all(obj[i + 1] < obj[i] for i in range(n_it))
    r0 = b.norm().item()
    r1 = (b - plan.matvec(y.dat, 0.9, 0.006)).norm().item()
    assert r1 < 0.2 * r0


# more channels than the joint-TV kernels take per launch (8): chained launches (_update.py:160-193
# loops over any number of channels)
MANY_CHANNELS = {
    'id_11ch': dict(dim_y=(10, 9, 21), n_channels=11, regime='id'),
    'dn_17ch': dict(dim_y=(9, 8, 11), n_channels=17, regime='dn', rot=0.05, trans=0.7),
}


@pytest.mark.parametrize('case', ['sr_3ch_axes', 'dn_2ch', 'id_2rep', 'sr_aniso', 'id_11ch', 'dn_17ch', 'sr_orient', 'dn_orient'])
@pytest.mark.parametrize('alpha', [1.0, 1.5])
def test_zw_update_and_objective_match_oracle(dev, case, alpha):
    """z/w updates and the objective, same inputs as the oracle."""
    import unires_amd as U
    prob = make_problem(seed=17, **{**CASES, **MANY_CHANNELS}[case])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    sett.alpha = alpha
    rho = torch.tensor(prob['rho'])
    zo, wo = prob['z'].clone() * 50, prob['w'].clone() * 50
    zg, wg = zo.clone().to(dev), wo.clone().to(dev)
    n_ref = O.compute_nll(xo, yo, prob['method'], prob['do_proj'])
    n_gpu = U._compute_nll(xg, yg, sett, float(rho))
    for a, b in zip(n_gpu, n_ref):
        assert abs(a.item() - b.item()) < 2e-5 * abs(b.item())
    zo, wo, tmp_o = O.update_zw(yo, zo, wo, rho, alpha=alpha)
    tmp = torch.zeros_like(yg[0].dat)
    zg, wg, tmp = U._update_zw(yg, zg, wg, float(rho), tmp, sett)
    assert rel_err(zg.cpu(), zo) < 2e-5 and rel_err(wg.cpu(), wo) < 2e-5
    assert rel_err(tmp.cpu(), tmp_o) < 2e-5


def test_full_admm_iterations_track_the_oracle(dev):
    """Three complete ADMM iterations (y, objective, z, w) against the oracle."""
    import unires_amd as U
    prob = make_problem(seed=18, **CASES['sr_3ch_axes'])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    zo, wo = prob['z'].clone(), prob['w'].clone()
    zg, wg = zo.clone().to(dev), wo.clone().to(dev)
    tmp = torch.zeros_like(yg[0].dat)
    obj = torch.zeros((3, 3), dtype=torch.float64, device=dev)
    for it in range(3):
        yo = O.update_y(xo, yo, zo, wo, rho, prob['method'], prob['do_proj'])
        ref_obj = O.compute_nll(xo, yo, prob['method'], prob['do_proj'])
        zo, wo, _ = O.update_zw(yo, zo, wo, rho)
        yg, zg, wg, tmp, obj = U._update_admm(xg, yg, zg, wg, float(rho), tmp, obj, it, sett)
        for c in range(3):
            assert rel_err(yg[c].dat.cpu(), yo[c].dat) < GATE, (it, c)
        assert abs(obj[it, 0].item() - ref_obj[0].item()) < 1e-4 * abs(ref_obj[0].item())
    assert rel_err(zg.cpu(), zo) < 5e-4 and rel_err(wg.cpu(), wo) < 5e-4


@pytest.mark.parametrize('case', ['sr_3ch_axes', 'sr_2rep', 'dn_2ch'])
def test_init_y_dat_matches_oracle(dev, case):
    """the initial trilinear reslice of the inputs into the mean space."""
    import unires_amd as U
    prob = make_problem(seed=19, **CASES[case])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    # the observations carry their own affine: mat_x composed with the rigid, as after coreg
    for c, ch in enumerate(prob['chans']):
        for n, r in enumerate(ch['reps']):
            m = (r['rigid'] @ r['mat_x'])
            xo[c][n].mat, xg[c][n].mat = m, m
    yo = O.init_y_dat(xo, yo)
    yg = U._init_y_dat(xg, yg, sett)
    for c in range(len(yo)):
        assert rel_err(yg[c].dat.cpu(), yo[c].dat) < 2e-5


@pytest.mark.parametrize('variant', ['tile', 'splat_short'])
def test_push_kernel_variants_match_oracle(dev, variant):
    """The alternative push kernels (UNIRES_PUSH=tile, or the short splat tile
    UNIRES_SPLAT_CFG=short; chosen at library load, default is the long-tile LDS splat) run
    the same parity gate in a fresh process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.helpers import make_problem, run_oracle_update_y, run_gpu_update_y, rel_err\n"
            "from tests.test_gpu_path import CASES\n"
            "for case in ('sr_3ch_axes', 'dn_2ch', 'sr_2rep'):\n"
            "    prob = make_problem(seed=21, **CASES[case])\n"
            "    yr, ir = run_oracle_update_y(prob)\n"
            "    yg, ig = run_gpu_update_y(prob, 'cuda:0')\n"
            "    for c in range(len(yr)):\n"
            "        assert ig[c][0] == ir[c][0]\n"
            "        assert rel_err(yg[c].cpu(), yr[c]) < 1e-4, case\n"
            "print('variant OK')\n") % root
    env = dict(os.environ, **({'UNIRES_SPLAT_CFG': 'short'} if variant == 'splat_short'
                              else {'UNIRES_PUSH': variant}))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and 'variant OK' in out.stdout, out.stderr[-2000:]


def test_aligned_kernel_agrees_with_general_kernels_at_256(dev, tmp_path):
    """Full-size cross-check of two independent implementations: config 3 with rigid = I runs
    the one-kernel aligned matvec; a fresh process with UNIRES_NO_ALIGNED=1 runs the general
    pull_conv + splat kernels on the same input.  Both also run 5 CG iterations."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "import unires_amd as U\n"
            "from unires_amd._project import _channel_plan\n"
            "dev = torch.device('cuda:0'); dim_y = (256, 256, 256)\n"
            "eye = torch.eye(4, dtype=torch.float64)\n"
            "D = torch.diag(torch.tensor([1, 1, 6, 1.], dtype=torch.float64))\n"
            "po = U._proj_info(dim_y, eye, (256, 256, 42), eye @ D, device=dev)\n"
            "g = torch.Generator().manual_seed(3)\n"
            "x = [U._input(torch.rand((256, 256, 42), generator=g).to(dev), eye @ D, 1.8e-4, po)]\n"
            "x[0].po.scl = 0.1\n"
            "y = U._output(torch.zeros(dim_y, device=dev), eye, 0.006)\n"
            "plan = _channel_plan(x, y, 'super-resolution', True)\n"
            "p = torch.rand(dim_y, generator=g).to(dev)\n"
            "q = plan.matvec(p, 0.9, 0.006)\n"
            "b = plan.rhs([x[0].dat], torch.zeros((3,) + dim_y, device=dev),\n"
            "             torch.zeros((3,) + dim_y, device=dev), 0.9, 0.006)\n"
            "n_it, obj = plan.cg(b, y.dat, 0.9, 0.006, max_iter=5, tolerance=1e-9)\n"
            "torch.save({'q': q.cpu(), 'y': y.dat.cpu(), 'obj': obj}, sys.argv[1])\n") % root
    res = {}
    for name, extra in (('aligned', {}), ('general', {'UNIRES_NO_ALIGNED': '1'})):
        path = str(tmp_path / (name + '.pt'))
        out = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, **extra),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[name] = torch.load(path)
    assert rel_err(res['aligned']['q'], res['general']['q']) < 2e-6
    assert rel_err(res['aligned']['y'], res['general']['y']) < 2e-5
    assert torch.allclose(torch.tensor(res['aligned']['obj']), torch.tensor(res['general']['obj']),
                          rtol=1e-6)


@pytest.mark.parametrize('case', ['sr_3ch_axes', 'sr_2rep', 'sr_aligned', 'sr_orient'])
@pytest.mark.parametrize('n_ls', [0, 4])
def test_update_scaling_matches_oracle(dev, case, n_ls):
    """SURVEY 8(f) next-3, even/odd slice-scaling Gauss-Newton (unires/_update.py:270-393):
    observations generated with a scaling the operator does not know yet."""
    import unires_amd as U
    prob = make_problem(seed=31, **CASES[case])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    g = torch.Generator().manual_seed(5)
    for c in range(len(xo)):
        yo[c].dat = torch.rand(prob['dim_y'], generator=g) * 100 + 20
        yg[c].dat = yo[c].dat.clone().to(dev)
        for n in range(len(xo[c])):
            po = xo[c][n].po
            true_scl = 0.12 if (c + n) % 2 == 0 else -0.08
            keep = po.scl
            po.scl = torch.tensor(true_scl)
            dat = O.proj_apply('A', yo[c].dat[None, None], po, method=prob['method'])[0, 0]
            po.scl = keep
            dat = dat + torch.randn(dat.shape, generator=g)
            dat[0, 0, :] = 0  # some masked-out voxels
            xo[c][n].dat = dat
            xg[c][n].dat = dat.clone().to(dev)
    xo, sll_o = O.update_scaling(xo, yo, method=prob['method'], max_niter_gn=2, num_linesearch=n_ls)
    xg, sll_g = U._update_scaling(xg, yg, sett, max_niter_gn=2, num_linesearch=n_ls)
    assert abs(sll_g.item() - sll_o.item()) < 2e-5 * abs(sll_o.item())
    for c in range(len(xo)):
        for n in range(len(xo[c])):
            so, sg = float(xo[c][n].po.scl), float(xg[c][n].po.scl)
            assert abs(sg - so) < 1e-5, (c, n, so, sg)
            assert abs(so - float(prob['chans'][c]['reps'][n].get('scl', 0.0))) > 1e-3  # it moved
    # the rebuilt operator uses the new scaling
    Ay_o = O.proj('A', yo[0].dat, xo[0], yo[0], method=prob['method'], do=True, n=0)
    Ay_g = U._proj('A', yg[0].dat, xg[0], yg[0], method=prob['method'], do=True, n=0)
    assert rel_err(Ay_g.cpu(), Ay_o) < 2e-5


def test_fit_loop_matches_oracle(dev):
    """The iteration loop of fit() (unires/run.py:56-143): regularisation schedule, rho,
    ADMM iterations, convergence countdowns, coarse-to-fine switch - GPU driver vs oracle."""
    import unires_amd as U
    prob = make_problem(seed=41, **CASES['sr_2rep'])
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    for c in range(len(yo)):
        yo[c].lam0 = yo[c].lam / 4.0
        yg[c].lam0 = float(yg[c].lam) / 4.0
    sett.max_iter, sett.tolerance, sett.reg_scl, sett.sched_num = 25, 1e-4, 4.0, 3
    sett.clean_fov = True
    y_ref, obj_ref, n_ref, sched = O.fit(xo, yo, prob['method'], prob['do_proj'], max_iter=25)
    y_pre = None
    dat, mat, R, info = U.fit(xg, yg, sett)
    assert info['n_iter'] == n_ref
    assert torch.equal(info['reg_scl'].cpu(), sched) and sched.tolist() == [32.0, 16.0, 8.0, 4.0]
    assert torch.allclose(info['obj'], obj_ref, rtol=5e-4)
    assert dat.shape == prob['dim_y'] + (len(yo),) and R.shape == (4, 4, 4)
    # clean_fov (run.py:150-164) restated with torch ops on the oracle result
    for c in range(len(yo)):
        keep = torch.ones(prob['dim_y'], dtype=torch.bool)
        for n, xn in enumerate(xo[c]):
            M = torch.linalg.solve(prob['mat_y'], xn.po.rigid.mm(xn.mat)).inverse()
            g = N.affine_grid(M.float(), prob['dim_y'])
            for d in range(3):
                keep &= (g[..., d] >= 0) & (g[..., d] < xn.dat.shape[d])
        ref = y_ref[c].dat.clone()
        ref[~keep] = 0
        assert (~keep).any()
        assert rel_err(dat[..., c].cpu(), ref) < 2e-3


def _rigid_setup(prob, dev, perturb=0.02):
    """Observations consistent with po.rigid; the registration then starts from a slightly
    wrong rigid (both sides express the same start matrix in their own se(3) basis)."""
    import unires_amd as U
    from unires_amd._rigid import _logq
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    Bo, Bg = O.affine_basis_se3(), U.affine_basis('SE')
    sett.rigid_basis = Bg
    g = torch.Generator().manual_seed(9)
    for c in range(len(xo)):
        yo[c].dat = prob['truth'][c].clone().float()
        yg[c].dat = yo[c].dat.clone().to(dev)
        for n in range(len(xo[c])):
            dq = (torch.rand(6, generator=g, dtype=torch.float64) * 2 - 1) * perturb
            dq[:3] *= 20  # translations in mm
            start = xo[c][n].po.rigid.mm(O.expm(dq, Bo))
            for xs, B in ((xo, Bo), (xg, Bg)):
                xs[c][n].rigid_q = _logq(start, B)
                xs[c][n].po.rigid = start.clone()
    return xo, yo, xg, yg, sett, Bo


@pytest.mark.parametrize('case', ['sr_3ch_axes', 'dn_2ch', 'sr_orient', 'dn_orient'])
def test_rigid_match_terms(dev, case):
    """_rigid_match (unires/_update.py:448-538): ll, gradient and Hessian volumes."""
    import unires_amd as U
    prob = make_problem(seed=51, **CASES[case])
    xo, yo, xg, yg, sett, Bo = _rigid_setup(prob, dev)
    method = prob['method']
    for c in range(len(xo)):
        po_o, po_g = xo[c][0].po, xg[c][0].po
        rigid = po_o.rigid
        CtC_o = CtC_g = None
        if method == 'super-resolution':
            import torch.nn.functional as F
            dim = tuple(po_o.dim_yx)
            CtC_o = F.conv_transpose3d(F.conv3d(torch.ones((1, 1) + dim), po_o.smo_ker, stride=po_o.ratio),
                                       po_o.smo_ker, stride=po_o.ratio)[0, 0]
            from unires_amd._rigid import _ctc
            CtC_g = _ctc(po_g, dim, dev)
            assert rel_err(CtC_g.cpu(), CtC_o) < 1e-5
        ll_o, gr_o, H_o = O.rigid_match(xo[c][0].dat, yo[c].dat[None, None], po_o, xo[c][0].tau, rigid,
                                        method, CtC=CtC_o, diff=True)
        ll_g, gr_g, H_g = U._rigid_match(xg[c][0].dat, yg[c].dat, po_g, xg[c][0].tau, rigid, sett,
                                         CtC=CtC_g, diff=True)
        assert abs(ll_g.item() - ll_o.item()) < 2e-5 * abs(ll_o.item())
        assert rel_err(gr_g.cpu(), gr_o) < 5e-5 and rel_err(H_g.cpu(), H_o) < 5e-5


@pytest.mark.parametrize('case', ['sr_3ch_axes', 'dn_2ch', 'sr_2rep', 'sr_orient', 'dn_orient'])
def test_update_rigid_matches_oracle(dev, case):
    """Unified rigid Gauss-Newton (unires/_update.py:198-266, 541-710): the oracle works in a
    different se(3) basis than the product - the rigid matrices and log-likelihoods agree."""
    import unires_amd as U
    prob = make_problem(seed=52, **CASES[case])
    xo, yo, xg, yg, sett, Bo = _rigid_setup(prob, dev)
    start = [[xn.po.rigid.clone() for xn in xc] for xc in xo]
    xo, sll_o = O.update_rigid(xo, yo, prob['method'], Bo, mean_correct=True, max_niter_gn=2,
                               num_linesearch=4)
    xg, sll_g = U._update_rigid(xg, yg, sett, mean_correct=True, max_niter_gn=2, num_linesearch=4,
                                samp=1)
    assert abs(sll_g.item() - sll_o.item()) < 1e-3 * abs(sll_o.item())
    moved = 0.0
    for c in range(len(xo)):
        for n in range(len(xo[c])):
            Ro, Rg = xo[c][n].po.rigid, xg[c][n].po.rigid
            assert (Rg - Ro).abs().max() < 2e-3 * max(1.0, Ro[:3, 3].abs().max().item()), (c, n)
            assert torch.allclose(U._expm(xg[c][n].rigid_q, sett.rigid_basis), Rg)
            moved = max(moved, (Ro - start[c][n]).abs().max().item())
    assert moved > 1e-3


def test_fit_with_rigid_and_scaling_updates(dev):
    """fit() with unified_rigid and scaling switched on (run.py:115-132): five iterations of
    ADMM + slice-scaling GN + rigid GN, GPU driver vs oracle."""
    import unires_amd as U
    prob = make_problem(seed=53, **CASES['sr_2rep'])
    xo, yo, xg, yg, sett, Bo = _rigid_setup(prob, dev, perturb=0.01)
    for c in range(len(yo)):
        yo[c].dat = prob['y0'][c].clone()
        yg[c].dat = prob['y0'][c].clone().to(dev)
        yo[c].lam0 = yo[c].lam / 4.0
        yg[c].lam0 = float(yg[c].lam) / 4.0
    sett.max_iter, sett.tolerance, sett.reg_scl, sett.sched_num = 5, 1e-4, 4.0, 3
    sett.unified_rigid, sett.scaling, sett.rigid_samp = True, True, 1
    y_ref, obj_ref, n_ref, sched = O.fit(xo, yo, prob['method'], prob['do_proj'], max_iter=5,
                                         scaling=True, unified_rigid=True, rigid_basis=Bo)
    dat, mat, R, info = U.fit(xg, yg, sett)
    assert info['n_iter'] == n_ref == 5
    assert torch.allclose(info['obj'], obj_ref, rtol=2e-3)
    k = 0
    for c in range(len(xo)):
        for n in range(len(xo[c])):
            assert (R[k] - xo[c][n].po.rigid).abs().max() < 5e-3
            assert abs(float(xg[c][n].po.scl) - float(xo[c][n].po.scl)) < 1e-4
            k += 1
        assert rel_err(dat[..., c].cpu(), y_ref[c].dat) < 5e-3


def test_fit_repeats_bit_for_bit(dev):
    """Two runs of fit() with the scaling and rigid updates on, from the same start: reconstruction, objective
    trace and scalings are identical to the last bit (every float64 reduction on the device - CG dot products,
    objective, likelihood, Gauss-Newton sums - adds per-workgroup sums in a fixed order; none uses atomics).
    The rigid parameters pass through the host's float64 6 x 6 solve and matrix exponential (LAPACK, whose
    last ulp may depend on buffer alignment): equal to 1e-12."""
    import unires_amd as U
    prob = make_problem(seed=54, **CASES['sr_3ch_axes'])
    outs = []
    for rep in range(2):
        xo, yo, xg, yg, sett, Bo = _rigid_setup(prob, dev, perturb=0.01)
        for c in range(len(yg)):
            yg[c].dat = prob['y0'][c].clone().to(dev)
            yg[c].lam0 = float(yg[c].lam) / 4.0
        sett.max_iter, sett.tolerance, sett.reg_scl, sett.sched_num = 4, 1e-4, 4.0, 3
        sett.unified_rigid, sett.scaling, sett.rigid_samp = True, True, 1
        dat, mat, R, info = U.fit(xg, yg, sett)
        outs.append((dat.clone(), info['obj'].clone(), [r.clone() for r in R],
                     [float(xn.po.scl) for xc in xg for xn in xc]))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[3] == b[3]
    assert all((p - q).abs().max() < 1e-12 for p, q in zip(a[2], b[2]))


@pytest.mark.parametrize('case', ['sr_3ch_axes', 'dn_2ch', 'sr_2rep'])
def test_update_y_on_channel_streams_and_one_after_the_other(dev, case):
    """settings.channel_streams: True / False / 'auto' run the same kernels per channel - streams only
    change what overlaps.  Both forms, run twice (the second call takes sum_n tau_n At x_n from the plan's
    cache, settings.cache_atx), give bit-identical y; the synchronous path the other tests compare with
    the oracle assembles its right-hand side in one kernel instead of cached + prior part, so it agrees
    to rounding."""
    import unires_amd as U
    from unires_amd._update import channel_streams_on
    prob = make_problem(**CASES[case])
    y_sync, _ = run_gpu_update_y(prob, dev, max_iter=6, tol=0.0)
    outs = {}
    for cs in (True, False, 'auto'):
        x, y, sett = gpu_structs(prob, dev)
        sett.cgs_max_iter, sett.cgs_tol, sett.channel_streams = 6, 0.0, cs
        assert channel_streams_on(sett, y[0].dat) == (cs is not False)  # (small volume: 'auto' = streams)
        z, w = prob['z'].to(dev), prob['w'].to(dev)
        tmp = torch.zeros_like(y[0].dat)
        for rep in range(2):
            for yc, y0 in zip(y, prob['y0']):
                yc.dat.copy_(y0.to(dev))
            U._update_y(x, y, z, w, prob['rho'], tmp, sett)
            torch.cuda.synchronize()
            outs[(cs, rep)] = [yc.dat.clone() for yc in y]
    first = outs[(True, 0)]
    for key, ys in outs.items():
        for c, yc in enumerate(ys):
            assert torch.equal(yc, first[c]), 'channel %d differs for channel_streams=%r, call %d' % ((c,) + key)
    for c, yc in enumerate(first):
        assert rel_err(yc.cpu(), y_sync[c].cpu()) < 1e-5


def test_matvec_timing_hooks_count_the_solve_and_leave_it_unchanged(dev):
    """unires_plan_time_matvecs / unires_plan_matvec_time (bench.py's roofline leg): one event pair per
    A(p) of the solve - max_iter of them in fixed-iteration mode - and the solve, then run as plain
    launches instead of a hipGraph, returns the same bits.  Mode 2 (every A(p) enqueued twice inside the solve's
    graph, bench.py's `us_per_launch_in_graph`): no events, the same bits again."""
    import unires_amd as U
    from unires_amd._project import _channel_plan
    prob = make_problem(**CASES['sr_3ch_axes'])
    outs = []
    for timed in (False, True, 2):
        x, y, sett = gpu_structs(prob, dev)
        sett.cgs_max_iter, sett.cgs_tol = 7, 0.0
        z, w = prob['z'].to(dev), prob['w'].to(dev)
        tmp = torch.zeros_like(y[0].dat)
        plans = [_channel_plan(x[c], y[c], sett.method, sett.do_proj) for c in range(len(x))]
        for pl in plans:
            pl.time_matvecs(timed)
        U._update_y(x, y, z, w, prob['rho'], tmp, sett)
        torch.cuda.synchronize()
        for pl in plans:
            n, us = pl.matvec_time()
            assert n == (7 if timed is True else 0)
            assert (us > 0.0) == (timed is True)
            pl.time_matvecs(False)
        outs.append([yc.dat.clone() for yc in y])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_one_kernel_conv_passes_are_bit_identical_to_the_separate_ones():
    """k_conv_ydown_xdownup2 / k_conv1d_downup2_m / k_conv_up_yz2 (A^T A and A^T of stride-2 profiles) against the
    separate marching passes they replace, in fresh processes (the switches are read once)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for env_extra in ({}, {'UNIRES_CONV_YX': '0'}, {'UNIRES_CONV_YX': '0', 'UNIRES_CONV_DOWNUP': '0'},
                      {'UNIRES_UPYZ_LDS': '0'},
                      {'UNIRES_CONV_YX': '0', 'UNIRES_CONV_DOWNUP': '0', 'UNIRES_UPYZ_LDS': '0'}):
        r = subprocess.run([sys.executable, os.path.join(here, '_conv_fused_probe.py')],
                           env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith('CONV ')])
    assert len(outs[0]) == 8
    for o in outs[1:]:
        assert o == outs[0]


@pytest.mark.parametrize('case', ['sr_3ch_axes', 'sr_iso2_gauss_v4', 'sr_iso2_rect_v4', 'sr_shift', 'sr_orient'])
def test_scaling_update_in_place_equals_a_fresh_plan(dev, case):
    """The scaling Gauss-Newton step changes po.scl only: unires_plan_set_repeat then rewrites the tables that
    carry S(scl) and keeps window plan and splat schedule.  A p, A^T x and A^T A p after two such updates must be
    those of a plan built from scratch with the final scaling, bit for bit."""
    import unires_amd as U
    from unires_amd._project import _channel_plan
    prob = make_problem(seed=77, **CASES[case])
    xg, yg, sett = gpu_structs(prob, dev)
    xf, yf, _ = gpu_structs(prob, dev)
    torch.manual_seed(3)
    for c in range(len(xg)):
        p = torch.rand(prob['dim_y'], device=dev) * 100
        _channel_plan(xg[c], yg[c], prob['method'], prob['do_proj'])  # built with the problem's scaling
        for step, d in enumerate((0.013, -0.021)):
            for xn in xg[c]:
                xn.po.scl = float(xn.po.scl) + d
            q = U._proj('AtA', p, xg[c], yg[c], method=prob['method'], do=prob['do_proj'], rho=torch.tensor(prob['rho']),
                        vx_y=torch.ones(3))
        for a, b in zip(xf[c], xg[c]):
            a.po.scl = float(b.po.scl)
        q_f = U._proj('AtA', p, xf[c], yf[c], method=prob['method'], do=prob['do_proj'], rho=torch.tensor(prob['rho']),
                      vx_y=torch.ones(3))
        assert torch.equal(q, q_f), (case, c)
        for n in range(len(xg[c])):
            xv = torch.rand(tuple(xg[c][n].po.dim_x), device=dev)
            for op, v in (('A', p), ('At', xv)):
                o1 = U._proj(op, v, xg[c], yg[c], method=prob['method'], do=prob['do_proj'], n=n)
                o2 = U._proj(op, v, xf[c], yf[c], method=prob['method'], do=prob['do_proj'], n=n)
                assert torch.equal(o1, o2), (case, c, n, op)
