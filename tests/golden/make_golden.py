"""Generates the golden fixtures in this directory FROM THE ORACLE (oracle/), because
nothing importable in the build container can run the reference itself (nitorch is
absent).  Each fixture stores seeded inputs and the oracle's outputs
for one small y-update problem, plus the nitorch semantics it assumes.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nitorch_restated as N  # noqa: E402
from oracle import unires_restated as O  # noqa: E402
from tests.helpers import make_problem, oracle_structs, run_oracle_update_y  # noqa: E402

CASES = {
    'sr_thick3_2ch': dict(dim_y=(12, 10, 9), n_channels=2, thick=3, regime='sr', scl=0.1, seed=101),
    'dn_rigid_1ch': dict(dim_y=(11, 9, 10), n_channels=1, regime='dn', rot=0.08, trans=1.2, seed=102),
    'id_2rep': dict(dim_y=(9, 8, 7), n_channels=1, regime='id', n_repeats=2, seed=103),
}
ASSUMED = dict(fov_tol=N.FOV_TOL, cg_stop='max_gain (objective 0.5*sum(x*(Ax-2b)))', cg_max_iter=20,
               cg_tol=1e-3, gauss_lim='floor((4w+2)/2)', nitorch_commit='8067d60 (recalled, absent)')


def build(name):
    prob = make_problem(**CASES[name])
    x, y = oracle_structs(prob)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    out = dict(assumed=np.array(repr(ASSUMED)), method=np.array(prob['method']),
               do_proj=np.array(prob['do_proj']), dim_y=np.array(prob['dim_y']),
               mat_y=prob['mat_y'].numpy(), rho=np.float32(prob['rho']), z=prob['z'].numpy(),
               w=prob['w'].numpy(), n_channels=np.array(len(x)))
    torch.manual_seed(5)
    for c in range(len(x)):
        out['lam_%d' % c] = np.float32(prob['chans'][c]['lam'])
        out['y0_%d' % c] = prob['y0'][c].numpy()
        out['n_rep_%d' % c] = np.array(len(x[c]))
        p = torch.rand(prob['dim_y']) * 100
        out['p_%d' % c] = p.numpy()
        out['AtAp_%d' % c] = O.proj('AtA', p, x[c], y[c], method=prob['method'], do=prob['do_proj'],
                                     rho=rho, vx_y=vx).numpy()
        out['b_%d' % c] = O.y_rhs(x[c], y[c], prob['z'][c], prob['w'][c], rho, vx, prob['method'],
                                  prob['do_proj']).numpy()
        for n, r in enumerate(prob['chans'][c]['reps']):
            k = '%d_%d' % (c, n)
            out['x_' + k] = r['dat'].numpy()
            out['mat_x_' + k] = r['mat_x'].numpy()
            out['rigid_' + k] = r['rigid'].numpy()
            out['tau_' + k] = np.float32(r['tau'])
            out['scl_' + k] = np.float32(r['scl'])
            out['dim_yx_' + k] = np.array(x[c][n].po.dim_yx)
            out['smo_ker_' + k] = x[c][n].po.smo_ker.numpy()
            out['Ay0_' + k] = O.proj('A', prob['y0'][c], x[c], y[c], method=prob['method'],
                                      do=prob['do_proj'], n=n).numpy()
    y_new, info = run_oracle_update_y(prob)
    for c in range(len(x)):
        out['y1_%d' % c] = y_new[c].numpy()
        out['cg_iters_%d' % c] = np.array(info[c][0])
        out['cg_obj_%d' % c] = info[c][1].numpy()
    return out


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        np.savez_compressed(os.path.join(here, name + '.npz'), **build(name))
        print('wrote', name)
