"""Which kernels the BASELINE configurations run on: every repeat of every configuration - at FULL
size, on the bench's own subjects - is served by the round-2+ kernels (the LDS-window pull, the schedule-driven
splat, the single-pass kernel of the denoising regime, the one-kernel translated form where it applies); none
falls through to the round-1 general kernels (`k_pull`, `k_pull_conv`, `k_splat`, `k_push_tile`), which remain
for operators outside those kernels' domains only (grids much finer or coarser than the output, tables beyond
their bit fields, more than 64 instructions per tile)."""
import pytest
import torch

import bench

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
    x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[name], dev, seed=1234)
    for c in range(len(x)):
        plan = _channel_plan(x[c], y[c], sett.method, sett.do_proj)
        for n in range(len(x[c])):
            info = plan.repeat_info(n)
            assert EXPECT[name](info), (name, c, n, info)
    del x, y, z, w
    torch.cuda.empty_cache()


def test_config1_is_the_identity_regime(dev):
    x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS['cfg1_181c1_denoise'], dev, seed=1234)
    assert sett.do_proj is False  # (A = I: the flat stencil kernel, no operator kernels at all)
