"""The single-pass A^T A kernel of the 'denoising' regime (unires_amd/csrc/ata1.hip): pull, push, stencil
and dot of `_proj('AtA')` (unires/_project.py:73-87, 180-188) in one kernel.  What the other GPU tests do
not pin: that the plans of the regime really take it, that it agrees with the pull + splat pair it replaces
(run in a fresh process with UNIRES_NO_ATA1=1), its three epilogue forms (store + dot, the CG objective,
accumulation over repeats), tiles with more segments than the LDS ring holds, volumes thinner than a tile
and every face of the volume in the field of view."""
import json
import os
import subprocess
import sys

import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import gpu_structs, make_problem, oracle_structs, rel_err

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    'dn_small': dict(dim_y=(15, 13, 11), n_channels=2, regime='dn', rot=0.1, trans=1.5),
    'dn_mid': dict(dim_y=(41, 38, 67), n_channels=1, regime='dn', rot=0.12, trans=3.0),
    'dn_2rep': dict(dim_y=(21, 19, 33), n_channels=1, regime='dn', n_repeats=2, rot=0.08, trans=2.0),
    # a strong rotation: rows cross a tile's thin cross-section within a few planes - many short segments per
    # tile (more than the 64 the LDS ring holds: the refill path)
    'dn_steep': dict(dim_y=(40, 44, 70), n_channels=1, regime='dn', rot=0.3, trans=2.0),
    # ... and one so steep that consecutive grid points share z planes all the time: the schedule's lane fill
    # drops below what the pull + splat pair does better, and the plan says so (fused False)
    'dn_steeper': dict(dim_y=(30, 34, 40), n_channels=1, regime='dn', rot=0.9, trans=2.0),
    'dn_thin': dict(dim_y=(3, 2, 5), n_channels=1, regime='dn', rot=0.05, trans=0.3),
    'dn_zoomed': dict(dim_y=(24, 22, 35), n_channels=1, regime='dn', rot=0.1, trans=1.0, vx_y=0.8),
}


def _matvec_both(dev, prob, seed=3):
    """(oracle, gpu, infos): _proj('AtA') with rho, every channel."""
    import unires_amd as U
    from unires_amd._project import _channel_plan
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    torch.manual_seed(seed)
    refs, outs, infos = [], [], []
    for c in range(len(xo)):
        p = torch.rand(prob['dim_y']) * 100
        refs.append(O.proj('AtA', p, xo[c], yo[c], method=prob['method'], rho=rho, vx_y=vx))
        outs.append(U._proj('AtA', p.to(dev), xg[c], yg[c], method=prob['method'], rho=rho, vx_y=vx).cpu())
        plan = _channel_plan(xg[c], yg[c], prob['method'], True, vx)
        infos.append([plan.repeat_info(n) for n in range(len(xg[c]))])
    return refs, outs, infos


@pytest.mark.parametrize('case', ['dn_small', 'dn_mid', 'dn_2rep', 'dn_steep', 'dn_zoomed'])
def test_denoising_plans_take_the_single_pass_kernel_and_match_the_oracle(dev, case):
    prob = make_problem(seed=11, **CASES[case])
    refs, outs, infos = _matvec_both(dev, prob)
    for c, (ref, out) in enumerate(zip(refs, outs)):
        assert rel_err(out, ref) < 2e-5, (case, c)
        for info in infos[c]:
            assert info['fused'], info  # none of these geometries is left to the pull + splat pair


def test_an_operator_the_schedule_fills_badly_is_left_to_the_pair(dev):
    prob = make_problem(seed=11, **CASES['dn_steeper'])
    refs, outs, infos = _matvec_both(dev, prob)
    assert rel_err(outs[0], refs[0]) < 2e-5
    assert not infos[0][0]['fused'], infos[0][0]


def test_volume_thinner_than_a_tile(dev):
    prob = make_problem(seed=5, **CASES['dn_thin'])
    refs, outs, _ = _matvec_both(dev, prob)
    assert rel_err(outs[0], refs[0]) < 2e-5


def test_cg_objective_and_multi_repeat_forms(dev):
    """The solve under the reference's stopping rule runs the kernel's objective form (A(x) folded into
    0.5 sum x (A x - 2 b), never stored), two repeats its accumulating form: iterate, iteration count and
    objective trace against the oracle."""
    from tests.helpers import run_gpu_update_y, run_oracle_update_y
    for case in ('dn_small', 'dn_2rep', 'dn_mid'):
        prob = make_problem(seed=21, **CASES[case])
        y_ref, info_ref = run_oracle_update_y(prob, max_iter=8, tol=1e-3)
        y_gpu, info_gpu = run_gpu_update_y(prob, dev, max_iter=8, tol=1e-3)
        for c in range(len(y_ref)):
            assert info_gpu[c][0] == info_ref[c][0], (case, c)
            assert rel_err(y_gpu[c].cpu(), y_ref[c]) < 1e-4, (case, c)


_CHILD = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
from tests.helpers import make_problem, gpu_structs
from oracle import nitorch_restated as N
import unires_amd as U
from unires_amd._project import _channel_plan
cases = %(cases)r
out = {}
for name, kw in cases.items():
    prob = make_problem(seed=11, **kw)
    xg, yg, sett = gpu_structs(prob, 'cuda:0')
    rho = torch.tensor(prob['rho']); vx = N.voxel_size(prob['mat_y']).float()
    torch.manual_seed(3)
    res = []
    for c in range(len(xg)):
        p = torch.rand(prob['dim_y']) * 100
        q = U._proj('AtA', p.to('cuda:0'), xg[c], yg[c], method=prob['method'], rho=rho, vx_y=vx).cpu()
        q2 = U._proj('AtA', p.to('cuda:0'), xg[c], yg[c], method=prob['method'], rho=rho, vx_y=vx).cpu()
        fused = [_channel_plan(xg[c], yg[c], prob['method'], True, vx).repeat_info(n)['fused'] for n in range(len(xg[c]))]
        res.append(dict(q=q.double().flatten().tolist(), same=bool(torch.equal(q, q2)), fused=fused))
    out[name] = res
json.dump(out, open(sys.argv[1], 'w'))
'''


def _child(tmp_path, tag, env_extra, cases):
    path = str(tmp_path / ('ata1_%s.json' % tag))
    env = dict(os.environ)
    env.update(env_extra)
    code = _CHILD % dict(root=ROOT, cases=cases)
    r = subprocess.run([sys.executable, '-c', code, path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.load(open(path)), r.stderr


def test_single_pass_kernel_against_the_pull_and_splat_pair(tmp_path):
    """Fresh processes: the default path (single pass) and UNIRES_NO_ATA1=1 (k_pull_conv2 + k_splat2, the r4
    path).  Same operator, different summation order: they agree to float32 rounding; each is bit-reproducible;
    and the steep case really exercises the ring refill (more than 64 segments in some tile)."""
    cases = {k: CASES[k] for k in ('dn_small', 'dn_2rep', 'dn_steep')}
    one, err_one = _child(tmp_path, 'one', {'UNIRES_ATA1_VERBOSE': '1'}, cases)
    two, _ = _child(tmp_path, 'two', {'UNIRES_NO_ATA1': '1'}, cases)
    for name in cases:
        for a, b in zip(one[name], two[name]):
            assert all(a['fused']) and not any(b['fused'])
            assert a['same'] and b['same']
            qa, qb = torch.tensor(a['q']), torch.tensor(b['q'])
            assert float((qa - qb).norm() / qb.norm()) < 2e-6, name
    segs = [int(l.rsplit(' ', 1)[1]) for l in err_one.splitlines() if 'max segments per tile' in l]
    assert segs and max(segs) > 64, segs
