"""Which kernels the BASELINE configurations run on: every repeat of every configuration - at FULL
size, on the bench's own subjects - is served by the round-2+ kernels (the LDS-window pull, the schedule-driven
splat, the single-pass kernel of the denoising regime, the one-kernel translated form where it applies); none
falls through to the round-1 general kernels (`k_pull`, `k_pull_conv`, `k_splat`, `k_push_tile`), which remain
for operators outside those kernels' domains only (grids much finer or coarser than the output, tables beyond
their bit fields, more than 64 instructions per tile)."""
import pytest
import torch

import workloads

pytestmark = pytest.mark.gpu

EXPECT = {
    # workload: predicate on repeat_info of every channel's repeat
    'cfg2_181c3_1mm': lambda i: i['fused'] and i['pull2'] and i['splat2_axis'] == -1,
    'cfg3_256c3_thick6z': lambda i: i['pull2'] and i['splat2_axis'] == 2 and not i['separable'],
    'cfg3_256c3_thick6_orient': lambda i: i['pull2'] and i['splat2_axis'] in (0, 1, 2) and not i['separable'],
    'cfg3_256c3_thick6xyz': lambda i: i['pull2'] and i['splat2_axis'] in (0, 1, 2) and not i['separable'],
    'cfg4_384c4_iso2': lambda i: i['pull2'] and i['splat2_axis'] == 2 and not i['separable'],
    'cfg4_384c4_iso2_gauss': lambda i: i['pull2'] and i['splat2_axis'] is not None,
    'demo_181c3_thick4xyz': lambda i: i['pull2'] and i['splat2_axis'] in (0, 1, 2) and not i['separable'],
}


@pytest.mark.parametrize('name', list(EXPECT))
def test_baseline_configurations_never_reach_the_round1_kernels(dev, name):
    from unires_amd._project import _channel_plan
    x, y, z, w, rho, sett = workloads.build_subject(workloads.WORKLOADS[name], dev, seed=1234)
    for c in range(len(x)):
        plan = _channel_plan(x[c], y[c], sett.method, sett.do_proj)
        for n in range(len(x[c])):
            info = plan.repeat_info(n)
            assert EXPECT[name](info), (name, c, n, info)
    del x, y, z, w
    torch.cuda.empty_cache()


def test_config1_is_the_identity_regime(dev):
    x, y, z, w, rho, sett = workloads.build_subject(workloads.WORKLOADS['cfg1_181c1_denoise'], dev, seed=1234)
    assert sett.do_proj is False  # (A = I: the flat stencil kernel, no operator kernels at all)


_BUILD_CHILD = r'''
import hashlib, json, sys, torch
sys.path.insert(0, %(root)r)
from tests.helpers import SIGNED_PERMS, make_problem, gpu_structs
from oracle import nitorch_restated as N
import unires_amd as U
cases = %(cases)r
out = {}
for name, kw in cases.items():
    if 'orient' in kw:
        kw = dict(kw, orient=[SIGNED_PERMS[i] for i in kw['orient']])
    prob = make_problem(seed=17, **kw)
    xg, yg, sett = gpu_structs(prob, 'cuda:0')
    rho = torch.tensor(prob['rho']); vx = N.voxel_size(prob['mat_y']).float()
    torch.manual_seed(5)
    hs = []
    for c in range(len(xg)):
        p = torch.rand(prob['dim_y']) * 100
        q = U._proj('AtA', p.to('cuda:0'), xg[c], yg[c], method=prob['method'], rho=rho, vx_y=vx).cpu()
        hs.append(hashlib.sha256(q.numpy().tobytes()).hexdigest())
    out[name] = hs
json.dump(out, open(sys.argv[1], 'w'))
'''


def test_one_pass_schedule_build_equals_the_two_pass_build(tmp_path):
    """The splat's schedule is built in ONE pass into per-tile staging slots and compacted (splat2.hip,
    k_splat2_build<2> + k_splat2_compact); UNIRES_S2_BUILD_2PASS=1 runs the count + fill passes of rounds 2-4.
    Same schedule, bit for bit - seen through bit-identical matvecs: thick slices along z, along x / y, all three
    axes (the extended entries), the Gaussian in-plane profile and the pull / push regime."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = {
        'z': dict(dim_y=(40, 36, 70), n_channels=2, thick=5, regime='sr', rot=0.15, trans=2.0, scl=0.05),
        'xy': dict(dim_y=(44, 40, 38), n_channels=2, thick=4, regime='sr', rot=0.1, trans=1.5, thick_axes=[0, 1]),
        'iso': dict(dim_y=(36, 34, 40), n_channels=1, thick=2, regime='sr', rot=0.1, trans=1.0, iso=(2, 2, 2), prof_ip=2),
        'dn_pair': dict(dim_y=(30, 34, 40), n_channels=1, regime='dn', rot=0.9, trans=2.0),
        'orient': dict(dim_y=(38, 30, 45), n_channels=2, thick=3, regime='sr', rot=0.12, trans=2.0, orient=[9, 22]),
    }
    res = {}
    for tag, extra in (('one', {}), ('two', {'UNIRES_S2_BUILD_2PASS': '1'})):
        path = str(tmp_path / ('build_%s.json' % tag))
        r = subprocess.run([sys.executable, '-c', _BUILD_CHILD % dict(root=root, cases=cases), path],
                           env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = json.load(open(path))
    assert res['one'] == res['two']
