"""k_pull_conv2's workgroup dispatch: the plan's table ({record, bi, bj, bc} per dispatch index, one 16-byte scalar
load, round 6) against the walk computed in the kernel (UNIRES_P2_WTAB=0, read once per process: fresh processes).
The table only reorders which workgroup takes which block and saves the walk's divisions: 'A' and the matvec are
bit-identical, for the z-only layout, the in-plane-thick (GEN) layouts and a grid that is not a whole number of
workgroups."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, sys, hashlib, torch
sys.path.insert(0, %(root)r)
from tests.helpers import make_problem, gpu_structs
import unires_amd as U
from oracle import nitorch_restated as N
from unires_amd._project import _channel_plan
out = {}
for name, kw in %(cases)r.items():
    prob = make_problem(seed=5, **kw)
    xg, yg, sett = gpu_structs(prob, 'cuda:0')
    rho = torch.tensor(prob['rho']); vx = N.voxel_size(prob['mat_y']).float()
    torch.manual_seed(2)
    res = []
    for c in range(len(xg)):
        p = (torch.rand(prob['dim_y']) * 10).to('cuda:0')
        a = U._proj('A', p, xg[c], yg[c], method=prob['method'], rho=rho, vx_y=vx)
        q = U._proj('AtA', p, xg[c], yg[c], method=prob['method'], rho=rho, vx_y=vx)
        torch.cuda.synchronize()
        info = [_channel_plan(xg[c], yg[c], prob['method'], True, vx).repeat_info(n) for n in range(len(xg[c]))]
        res.append(dict(a=[hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest() for t in a],
                        q=hashlib.sha1(q.cpu().numpy().tobytes()).hexdigest(), pull2=[int(i['pull2']) for i in info]))
    out[name] = res
json.dump(out, open(sys.argv[1], 'w'))
'''

CASES = {
    'thick_z': dict(dim_y=(72, 66, 61), n_channels=2, thick=6, rot=0.1, trans=3.0),
    'thick_xyz': dict(dim_y=(64, 70, 58), n_channels=3, thick=4, thick_axes=[0, 1, 2], rot=0.08, trans=2.0),
    'dn_ragged': dict(dim_y=(45, 51, 70), n_channels=1, regime='dn', rot=0.1, trans=2.5),
}


def _child(tmp_path, tag, env_extra):
    path = str(tmp_path / ('dispatch_%s.json' % tag))
    env = dict(os.environ)
    env.update(env_extra)
    env['UNIRES_NO_ATA1'] = '1'  # the denoising case through the pull + splat pair
    r = subprocess.run([sys.executable, '-c', _CHILD % dict(root=ROOT, cases=CASES), path], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.load(open(path))


def test_dispatch_table_and_in_kernel_walk_give_the_same_bits(tmp_path):
    tab = _child(tmp_path, 'tab', {})
    walk = _child(tmp_path, 'walk', {'UNIRES_P2_WTAB': '0'})
    for name in CASES:
        assert len(tab[name]) == len(walk[name])
        for a, b in zip(tab[name], walk[name]):
            assert all(a['pull2']) and all(b['pull2']), (name, a['pull2'])  # the window kernel really ran
            assert a['a'] == b['a'], name
            assert a['q'] == b['q'], name
