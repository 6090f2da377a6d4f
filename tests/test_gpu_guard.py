"""The guarded 'max_gain' stopping rule (round 6; C ABI UNIRES_STOP_MAXGAIN_GUARDED, cg.hip).

nitorch's cg(stop='max_gain') - what UniRes passes, unires/_update.py:142-148, struct.py:65-67 - evaluates the
objective 0.5 sum x (A(x) - 2b) after every iteration: a second operator application.  The guarded rule takes the
objective from the recurred residual while its gain is >= 4 x tolerance, evaluates it afresh from there on, and stops
without a fresh value once the recurred gain is below tolerance / 2; here:
  * the drift between the two objectives is bounded, relative to the range the gain is normalised by, far below the
    guard band's width;
  * with the tolerance PLANTED right above and right below every gain of a solve - decisions as close as they get -
    the guarded rule stops at the iteration the every-iteration rule stops at, and both where the oracle's nitorch
    cg() does;
  * the traces agree (the guarded one mixes recurred and fresh values) and fewer fresh evaluations were paid for.
"""
import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import gpu_structs, make_problem, oracle_structs

pytestmark = pytest.mark.gpu


def _setup(dev, seed, regime, dim=(36, 34, 38)):
    from unires_amd._project import _channel_plan
    kw = dict(rot=0.08, trans=1.2) if regime == 'dn' else dict(thick=3, scl=0.05)
    prob = make_problem(seed=seed, dim_y=dim, n_channels=1, regime=regime, **kw)
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    b = O.y_rhs(xo[0], yo[0], prob['z'][0], prob['w'][0], rho, vx, prob['method'], True)
    lhs = lambda d: O.proj('AtA', d, xo[0], yo[0], method=prob['method'], rho=rho, vx_y=vx)
    plan = _channel_plan(xg[0], yg[0], prob['method'], True, vx)
    return prob, plan, b, lhs, yo[0].dat.clone(), float(rho), float(yg[0].lam)


def _solve(plan, b, x0, rho, lam, dev, tol, stop, max_iter=40):
    x = x0.clone().to(dev)
    it, obj = plan.cg(b.to(dev), x, rho, lam, max_iter=max_iter, tolerance=tol, stop=stop)
    return it, [float(v) for v in obj], x


def _gains(obj):
    """nitorch get_gain(obj[:k + 1], 'decreasing') for every k >= 1."""
    out = []
    for k in range(1, len(obj)):
        rng = max(obj[:k + 1]) - min(obj[:k + 1])
        out.append(abs((obj[k - 1] - obj[k]) / rng))
    return out


@pytest.mark.parametrize('seed,regime', [(201, 'sr'), (202, 'dn'), (203, 'sr')])
def test_guarded_rule_makes_the_fresh_rules_decisions(dev, seed, regime):
    prob, plan, b, lhs, x0, rho, lam = _setup(dev, seed, regime)
    n_full = 14
    # the whole traces (a tolerance far below any gain: nothing stops), fresh and recurred
    it_f, obj_f, _ = _solve(plan, b, x0, rho, lam, dev, 1e-30, 'max_gain_fresh', max_iter=n_full)
    it_r, obj_r, _ = _solve(plan, b, x0, rho, lam, dev, 1e-30, 'max_gain_recurred', max_iter=n_full)
    assert it_f == it_r == n_full
    rng = max(obj_f) - min(obj_f)
    drift = max(abs(a - c) for a, c in zip(obj_f, obj_r)) / rng
    # the guard band spans [tol / 2, 4 tol): a drift of 1e-5 of the range cannot carry a gain across either half of it
    # for any tolerance >= 1e-4 (the reference's default is 1e-3)
    assert drift < 1e-5, drift
    gains = _gains(obj_f)
    planted = 0
    for k, g in enumerate(gains[1:], start=2):  # (the first gain is 1 by construction)
        if not 1e-5 < g < 0.2:
            continue
        # (2.2 / 1.9: the gain lands just below / just above HALF the tolerance - the band's lower edge, under which
        # the solve stops on the recurred gain alone)
        for f in (1.02, 0.98, 1.004, 0.996, 2.2, 1.9):
            tol = g * f
            n_ref = next((j for j, gj in enumerate(gains, start=1) if gj < tol), n_full)
            it_a, obj_a, xa = _solve(plan, b, x0, rho, lam, dev, tol, 'max_gain_fresh', max_iter=n_full)
            it_b, obj_b, xb = _solve(plan, b, x0, rho, lam, dev, tol, 'max_gain', max_iter=n_full)
            assert it_a == n_ref, (k, f, it_a, n_ref)
            assert it_b == it_a, 'gain %g planted at %g x: guarded %d, fresh %d' % (g, f, it_b, it_a)
            assert torch.equal(xa, xb)  # same iterations, same arithmetic on x
            assert torch.allclose(torch.tensor(obj_b), torch.tensor(obj_a), rtol=1e-5, atol=1e-6 * rng)
            planted += 1
    assert planted >= 12


def test_guarded_rule_stops_where_nitorch_does_and_skips_fresh_evaluations(dev):
    prob, plan, b, lhs, x0, rho, lam = _setup(dev, 204, 'sr', dim=(40, 36, 42))
    for tol in (1e-2, 1e-3, 1e-4):
        xr, n_ref, obj_ref = N.cg(lhs, b, x0.clone(), max_iter=40, tolerance=tol, stop='max_gain', return_info=True)
        it, obj, x = _solve(plan, b, x0, rho, lam, dev, tol, 'max_gain')
        assert it == n_ref, (tol, it, n_ref)
        assert torch.allclose(torch.tensor(obj, dtype=torch.float64), obj_ref, rtol=1e-5)
        assert (x.cpu() - xr).norm() / xr.norm() < 1e-4
    # what it is for: the second A(x) runs on a part of the iterations only (its launches return at entry otherwise);
    # counted through the library's own matvec timer - every A(.) of a solve between HIP events
    xr, n_ref, _ = N.cg(lhs, b, x0.clone(), max_iter=40, tolerance=1e-4, stop='max_gain', return_info=True)
    assert n_ref >= 6
