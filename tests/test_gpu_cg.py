"""The chunked CG enqueue (round 4): solves that can stop early (tolerance > 0 - the reference's default,
struct.py:65-67) are fed to the device chunk by chunk from a host loop that watches a host-mapped
progress word.  Parity with nitorch's cg() as the oracle restates it is covered by every tol = 1e-3 case
of tests/test_gpu_path.py; here: the enqueue strategy changes nothing, the channels' joint solve equals the
channel-by-channel one, iteration budgets beyond the old 4 096-iteration graph work."""
import os
import subprocess
import sys

import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import gpu_structs, make_problem, oracle_structs, rel_err, run_oracle_update_y

pytestmark = pytest.mark.gpu


def test_chunked_enqueue_is_bit_identical_to_the_full_enqueue():
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for env_extra in ({'UNIRES_CG_CHUNK': '0'}, {}, {'UNIRES_CG_CHUNK': '1'}, {'UNIRES_CG_CHUNK': '3'},
                      {'UNIRES_CG_CHUNK': '7'}, {'UNIRES_CG_GRAPH': '0'}, {'UNIRES_CG_CHUNK': '0', 'UNIRES_CG_GRAPH': '0'}):
        r = subprocess.run([sys.executable, os.path.join(here, '_cg_chunk_probe.py')],
                           env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith('CG ')])
    assert len(outs[0]) == 18
    for o in outs[1:]:
        assert o == outs[0]
    # the problems do stop early (otherwise nothing was tested)
    assert any(int(l.split()[4]) < 20 for l in outs[0])


@pytest.mark.parametrize('streams', [True, False])
def test_joint_solve_of_the_channels_matches_oracle(dev, streams):
    """_update_y with the channels on separate streams (unires_cg_solve_many) and one after the other."""
    import unires_amd as U
    prob = make_problem(seed=71, dim_y=(20, 18, 16), n_channels=3, thick=4, regime='sr', scl=0.1)
    y_ref, info_ref = run_oracle_update_y(prob, max_iter=20, tol=1e-3)
    x, y, sett = gpu_structs(prob, dev)
    sett.cgs_max_iter, sett.cgs_tol, sett.channel_streams = 20, 1e-3, streams
    z, w = prob['z'].to(dev), prob['w'].to(dev)
    for rep in range(3):  # capture, replay, replay
        for c in range(3):
            y[c].dat = prob['y0'][c].clone().to(dev)
        U._update_y(x, y, z, w, prob['rho'], torch.zeros_like(y[0].dat), sett)
        torch.cuda.synchronize()
        for c in range(3):
            assert rel_err(y[c].dat.cpu(), y_ref[c]) < 1e-4, (rep, c)


def test_cg_many_returns_counts_and_traces(dev):
    from unires_amd._plan import cg_many
    from unires_amd._project import _channel_plan
    prob = make_problem(seed=72, dim_y=(15, 13, 11), n_channels=2, regime='dn', rot=0.1, trans=1.5)
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    plans, bs, refs = [], [], []
    for c in range(2):
        b = O.y_rhs(xo[c], yo[c], prob['z'][c], prob['w'][c], rho, vx, prob['method'], True)
        lhs = lambda d, c=c: O.proj('AtA', d, xo[c], yo[c], method=prob['method'], rho=rho, vx_y=vx)
        refs.append(N.cg(lhs, b, yo[c].dat.clone(), max_iter=20, tolerance=1e-3, stop='max_gain', return_info=True))
        plans.append(_channel_plan(xg[c], yg[c], prob['method'], True, vx))
        bs.append(b.to(dev))
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    res = cg_many(plans, bs, [yc.dat for yc in yg], float(rho), [float(yc.lam) for yc in yg], streams,
                  max_iter=20, tolerance=1e-3, sync=True)
    for c in range(2):
        xr, n_ref, obj_ref = refs[c]
        assert res[c][0] == n_ref and rel_err(yg[c].dat.cpu(), xr) < 1e-4
        assert torch.allclose(torch.tensor(res[c][1], dtype=torch.float64), obj_ref, rtol=1e-5)
    with pytest.raises(ValueError, match='one plan per solve'):
        cg_many([plans[0], plans[0]], bs, [yc.dat for yc in yg], float(rho), [1.0, 1.0], streams)


def test_iteration_budget_beyond_the_old_graph_limit(dev):
    """nitorch's cg() defaults to max_iter = 10 numel; rounds 1-3 capped the budget at the 4 096 iterations
    one captured graph held.  With a tolerance the solve is fed in chunks: any budget, the same answer."""
    from unires_amd import optim
    from unires_amd._project import _channel_plan
    prob = make_problem(seed=73, dim_y=(12, 11, 10), n_channels=1, regime='dn', rot=0.05, trans=0.7)
    xo, yo = oracle_structs(prob)
    xg, yg, sett = gpu_structs(prob, dev)
    rho = torch.tensor(prob['rho'])
    vx = N.voxel_size(prob['mat_y']).float()
    b = O.y_rhs(xo[0], yo[0], prob['z'][0], prob['w'][0], rho, vx, prob['method'], True)
    lhs = lambda d: O.proj('AtA', d, xo[0], yo[0], method=prob['method'], rho=rho, vx_y=vx)
    xr, n_ref, obj_ref = N.cg(lhs, b, yo[0].dat.clone(), max_iter=10 * b.numel(), tolerance=1e-5, stop='max_gain',
                              return_info=True)
    plan = _channel_plan(xg[0], yg[0], prob['method'], True, vx)
    n_gpu, obj = plan.cg(b.to(dev), yg[0].dat, float(rho), float(yg[0].lam), max_iter=10 * b.numel(),
                         tolerance=1e-5)
    assert n_gpu == n_ref and 2 < n_gpu < 4096
    assert rel_err(yg[0].dat.cpu(), xr) < 1e-4
    with pytest.raises(ValueError, match='needs a tolerance'):
        plan.cg(b.to(dev), yg[0].dat, float(rho), float(yg[0].lam), max_iter=5000, tolerance=0.0)
    # ... and under stream capture nothing runs until the graph is launched - the chunk feeder would wait for ever
    # and its watchdog would query a capturing stream: refused up front, the capture survives
    side = torch.cuda.Stream()
    bg = b.to(dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        g.capture_begin()
        try:
            with pytest.raises(Exception, match='cannot join a stream capture'):
                plan.cg(bg, yg[0].dat, float(rho), float(yg[0].lam), max_iter=5000, tolerance=1e-5, sync=False)
            yg[0].dat.mul_(1.0)  # (something to capture)
        finally:
            g.capture_end()
    g.replay()
    torch.cuda.synchronize()
