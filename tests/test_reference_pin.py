"""The UniRes half of the oracle, pinned against the reference's own code.

tests/golden/ref_*.npz were written by tests/golden/make_golden_from_reference.py, which imports
/root/reference/unires/_project.py and _update.py themselves (nitorch bound to
oracle/nitorch_restated) and runs _proj_info / _proj_apply / _proj / _update_admm /
_compute_nll / _update_scaling / _update_rigid_channel.  Here oracle/unires_restated.py has to
reproduce every one of those outputs from the same inputs - float32 round-off only (1e-6).
Parity stays unpinned at the nitorch boundary (nitorch is not in the build container).
"""
import os

import numpy as np
import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
# (vectors regenerated against the REAL nitorch - make_golden_from_reference.py --real-nitorch - live elsewhere)
GOLD_REF = os.environ.get('UNIRES_GOLDEN_DIR', GOLD)
CASES = ['ref_sr_2ch', 'ref_sr_2rep', 'ref_dn_1ch', 'ref_id_2rep', 'ref_sr_fine', 'ref_sr_gauss',
         'ref_sr_fine_gauss', 'ref_sr_orient', 'ref_sr_gauss_v4']
TOL = 1e-6


def load_case(name):
    g = np.load(os.path.join(GOLD_REF, name + '.npz'), allow_pickle=False)
    method, do_proj = str(g['method']), bool(g['do_proj'])
    dim_y = tuple(int(v) for v in g['dim_y'])
    mat_y = torch.from_numpy(g['mat_y'])
    x, y = [], []
    for c in range(int(g['n_channels'])):
        xc = []
        for n in range(int(g['n_rep_%d' % c])):
            k = '%d_%d' % (c, n)
            dat = torch.from_numpy(g['x_' + k])
            mat_x = torch.from_numpy(g['mat_x_' + k])
            po = O.proj_info(dim_y, mat_y, tuple(dat.shape), mat_x, rigid=torch.from_numpy(g['rigid_' + k]),
                             prof_ip=int(g['prof_ip']) if 'prof_ip' in g else 0, prof_tp=0,
                             scl=float(g['scl_' + k]))
            xn = O.make_input(dat.clone(), mat_x, torch.tensor(float(g['tau_' + k])), po)
            xn.rigid_q = torch.from_numpy(g['rigid_q_' + k]).clone()
            xc.append(xn)
        x.append(xc)
        y.append(O.make_output(torch.from_numpy(g['y0_%d' % c]).clone(), mat_y, torch.tensor(float(g['lam_%d' % c]))))
    return g, method, do_proj, dim_y, x, y


@pytest.mark.parametrize('name', CASES)
def test_proj_info_and_operators(name):
    """_proj_info (:193-297), _proj_apply (:99-190), _proj (:54-96)."""
    g, method, do_proj, dim_y, x, y = load_case(name)
    rho = torch.tensor(float(g['rho']))
    vx = N.voxel_size(y[0].mat).float()
    for c in range(len(x)):
        p = torch.from_numpy(g['p_%d' % c])
        assert rel_err(O.proj('AtA', p, x[c], y[c], method=method, do=do_proj, rho=rho, vx_y=vx),
                       torch.from_numpy(g['AtAp_%d' % c])) < TOL
        for n, xn in enumerate(x[c]):
            k = '%d_%d' % (c, n)
            po = xn.po
            assert tuple(int(v) for v in po.dim_yx) == tuple(int(v) for v in g['po_dim_yx_' + k])
            assert tuple(int(v) for v in po.ratio) == tuple(int(v) for v in g['po_ratio_' + k])
            assert int(po.dim_thick) == int(g['po_dim_thick_' + k])
            np.testing.assert_allclose(po.mat_yx.numpy(), g['po_mat_yx_' + k], rtol=0, atol=1e-12)
            np.testing.assert_allclose(po.smo_ker.numpy(), g['po_smo_ker_' + k], rtol=0, atol=1e-7)
            assert rel_err(O.proj('A', y[c].dat, x[c], y[c], method=method, do=do_proj, n=n),
                           torch.from_numpy(g['Ay0_' + k])) < TOL
            assert rel_err(O.proj('At', xn.dat, x[c], y[c], method=method, do=do_proj, n=n),
                           torch.from_numpy(g['Atx_' + k])) < TOL
            if do_proj:
                assert rel_err(O.proj_apply('AtA', p[None, None], po, method=method)[0, 0],
                               torch.from_numpy(g['AtA1_' + k])) < TOL


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('alpha,tag', [(1.0, ''), (1.5, '_a15')])
def test_admm_iteration(name, alpha, tag):
    """_update_admm (:105-195): y-update, objective, z- and w-update; _compute_nll (:396-427)."""
    g, method, do_proj, dim_y, x, y = load_case(name)
    rho = torch.tensor(float(g['rho']))
    z, w = torch.from_numpy(g['z']).clone(), torch.from_numpy(g['w']).clone()
    nll0 = O.compute_nll(x, y, method, do_proj)
    np.testing.assert_allclose([float(v) for v in nll0], g['nll0'], rtol=1e-9)
    y = O.update_y(x, y, z, w, rho, method, do_proj, cgs_max_iter=20, cgs_tol=1e-3)
    for c in range(len(y)):
        assert rel_err(y[c].dat, torch.from_numpy(g['y1%s_%d' % (tag, c)])) < TOL
    obj = O.compute_nll(x, y, method, do_proj)
    np.testing.assert_allclose([float(v) for v in obj], g['obj1' + tag], rtol=1e-7)
    z, w, _ = O.update_zw(y, z, w, rho, alpha=alpha)
    assert rel_err(z, torch.from_numpy(g['z1' + tag])) < TOL
    assert rel_err(w, torch.from_numpy(g['w1' + tag])) < TOL


def _after_admm(g, x, y, method, do_proj, tag='_a15'):
    for c in range(len(y)):
        y[c].dat = torch.from_numpy(g['y1%s_%d' % (tag, c)]).clone()
    return y


@pytest.mark.parametrize('name', ['ref_sr_2ch', 'ref_sr_2rep', 'ref_sr_fine', 'ref_sr_gauss', 'ref_sr_fine_gauss',
                                  'ref_sr_orient', 'ref_sr_gauss_v4'])
def test_update_scaling(name):
    """_update_scaling (:270-393): two Gauss-Newton iterations with line search."""
    g, method, do_proj, dim_y, x, y = load_case(name)
    y = _after_admm(g, x, y, method, do_proj)
    x, sll = O.update_scaling(x, y, method=method, max_niter_gn=2, num_linesearch=4)
    np.testing.assert_allclose(float(sll), float(g['scl_sll']), rtol=1e-7)
    for c in range(len(x)):
        for n, xn in enumerate(x[c]):
            np.testing.assert_allclose(float(xn.po.scl), float(g['scl1_%d_%d' % (c, n)]), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('name', ['ref_sr_2ch', 'ref_sr_2rep', 'ref_dn_1ch', 'ref_sr_fine', 'ref_sr_gauss',
                                  'ref_sr_fine_gauss', 'ref_sr_orient', 'ref_sr_gauss_v4'])
def test_update_rigid_channel(name):
    """_update_rigid_channel (:541-710), no sub-sampling, same se(3) basis as the fixture."""
    g, method, do_proj, dim_y, x, y = load_case(name)
    y = _after_admm(g, x, y, method, do_proj)
    basis = torch.from_numpy(g['basis'])
    for c in range(len(x)):
        xc, sll = O.update_rigid_channel(x[c], y[c], method, basis, max_niter_gn=1, num_linesearch=4,
                                         samp=int(g['rigid_samp']),
                                         prof_ip=int(g['prof_ip']) if 'prof_ip' in g else 0)
        np.testing.assert_allclose(float(sll), float(g['rig_sll_%d' % c]), rtol=1e-6)
        for n, xn in enumerate(xc):
            np.testing.assert_allclose(xn.rigid_q.numpy(), g['rig_q1_%d_%d' % (c, n)], rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(xn.po.rigid.numpy(), g['rig_rigid1_%d_%d' % (c, n)], rtol=0, atol=1e-8)
