"""GPU parity of every op-level entry point of the C ABI against the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import rel_err, rigid_matrix

pytestmark = pytest.mark.gpu

TOL = 1e-5  # float32 op-level agreement (relative L2); the path-level gate is 1e-4


def _affines():
    eye = torch.eye(4, dtype=torch.float64)
    shift = eye.clone()
    shift[:3, 3] = torch.tensor([1.0, -2.0, 3.0])
    scale = torch.diag(torch.tensor([1.0, 1.0, 3.0, 1.0], dtype=torch.float64))
    scale[2, 3] = -1.0
    return {'identity': eye, 'int_shift': shift, 'thick': scale,
            'small_rigid': rigid_matrix([0.4, -0.3, 0.2], [0.03, -0.02, 0.05]),
            'big_rigid': rigid_matrix([3.0, -4.0, 2.5], [0.4, -0.3, 0.6]),
            'far_outside': rigid_matrix([100.0, 0, 0], [0, 0, 0])}


@pytest.mark.parametrize('name', list(_affines()))
@pytest.mark.parametrize('sdim,gdim', [((12, 10, 9), (11, 12, 10)), ((5, 70, 131), (6, 66, 140)),
                                        ((1, 1, 1), (2, 3, 4))])
def test_pull_and_push(dev, name, sdim, gdim):
    from unires_amd import spatial
    torch.manual_seed(0)
    M = _affines()[name]
    src = torch.rand((1, 1) + sdim)
    g = N.affine_grid(M.float(), gdim)[None]
    ref = N.grid_pull(src, g)
    out = spatial.grid_pull(src.to(dev), M, gdim).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
    val = torch.rand((1, 1) + gdim)
    refp = N.grid_push(val, g, sdim)
    outp = spatial.grid_push(val.to(dev), M, sdim).cpu()
    assert outp.shape == refp.shape
    assert (outp - refp).abs().max() <= 2e-5 * max(1.0, refp.abs().max().item())


@pytest.mark.parametrize('name', list(_affines()))
@pytest.mark.parametrize('sdim,gdim', [((12, 10, 9), (11, 12, 10)), ((5, 70, 131), (6, 66, 140))])
def test_grid_grad(dev, name, sdim, gdim):
    """nitorch grid_grad (spatial derivatives of the trilinear pull; unires/_update.py:508):
    HIP vs the oracle, and the oracle vs a central finite difference of grid_pull."""
    from unires_amd import spatial
    torch.manual_seed(1)
    M = _affines()[name]
    src = torch.rand((1, 1) + sdim)
    g = N.affine_grid(M.float(), gdim)[None]
    ref = N.grid_grad(src, g)
    out = spatial.grid_grad(src.to(dev), M, gdim).cpu()
    assert out.shape == ref.shape == (1, 1) + gdim + (3,)
    assert (out - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
    if name == 'small_rigid':  # oracle pin: derivative of its own grid_pull away from cell borders
        h = 1e-2
        frac = g - g.floor()
        inner = ((frac > 0.05) & (frac < 0.95)).all(-1) & (N.grid_pull(torch.ones_like(src), g)[0, 0] > 0.999)
        for d in range(3):
            e = torch.zeros(3)
            e[d] = h
            gd_, ed_ = g.double(), e.double()
            fd = (N.grid_pull(src.double(), gd_ + ed_) - N.grid_pull(src.double(), gd_ - ed_)) / (2 * h)
            assert (fd[0, 0][inner[0]] - ref[0, 0, ..., d][inner[0]]).abs().max() < 2e-5


@pytest.mark.parametrize('zscale', [0.9, 0.45, 0.12])
def test_push_with_a_grid_much_finer_than_the_output(dev, zscale):
    """Long grid rows per output tile (5+ segments per row): the segment list of the tile
    kernels is filled in several passes - nothing may be dropped."""
    from unires_amd import spatial
    torch.manual_seed(11)
    sdim, gdim = (20, 12, 40), (22, 14, int(40 / zscale) + 3)
    M = rigid_matrix([0.3, -0.2, 0.4], [0.02, -0.03, 0.01])
    M[:3, 2] *= zscale
    val = torch.rand((1, 1) + gdim)
    g = N.affine_grid(M.float(), gdim)[None]
    ref = N.grid_push(val, g, sdim)
    out = spatial.grid_push(val.to(dev), M, sdim).cpu()
    assert (out - ref).abs().max() <= 5e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('rot', [(0, 0, 0), (0.02, -0.01, 0.03), (0.1, 0.1, -0.1), (0.0, 0.25, 0.0),
                                 (0.3, 0.0, 0.0), (-0.2, 0.15, 0.5), (0.9, 0.0, 0.0)])
@pytest.mark.parametrize('scale', [1.0, 0.93, 1.21])
def test_push_is_race_free_and_reproducible(dev, rot, scale):
    """The splat kernel updates its LDS tile without atomics; every geometry class of
    its safety argument (paired rows, same-plane neighbours, atomics fallback) is hit
    here on a volume spanning many tiles, and two runs must agree bit for bit."""
    from unires_amd import spatial
    torch.manual_seed(7)
    sdim, gdim = (37, 29, 95), (40, 33, 101)
    M = rigid_matrix([1.7, -2.2, 0.9], rot)
    M[:3, :3] *= scale
    val = torch.rand((1, 1) + gdim)
    g = N.affine_grid(M.float(), gdim)[None]
    ref = N.grid_push(val, g, sdim)
    a = spatial.grid_push(val.to(dev), M, sdim)
    b = spatial.grid_push(val.to(dev), M, sdim)
    assert torch.equal(a, b)
    assert (a.cpu() - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
    src = torch.rand((1, 1) + sdim)
    refp = N.grid_pull(src, g)
    outp = spatial.grid_pull(src.to(dev), M, gdim).cpu()
    assert (outp - refp).abs().max() <= 2e-5


def test_push_accumulates_with_alpha(dev):
    from unires_amd import _ops, spatial
    torch.manual_seed(1)
    M = _affines()['small_rigid']
    val = torch.rand(9, 8, 7)
    base = torch.rand(10, 9, 8)
    ref = base + 0.37 * N.grid_push(val[None, None], N.affine_grid(M.float(), (9, 8, 7))[None],
                                    (10, 9, 8))[0, 0]
    out = base.to(dev)
    _ops.push_affine(val.to(dev), spatial._m12(M), (10, 9, 8), alpha=0.37, out=out)
    assert rel_err(out.cpu(), ref) < TOL


@pytest.mark.parametrize('stride,kinds,w', [((1, 1, 3), (-1, -1, 0), 3.0), ((4, 1, 1), (0, -1, -1), 4.0),
                                             ((1, 6, 1), (-1, 0, -1), 6.0), ((2, 2, 2), (0, 2, 2), 2.0),
                                             ((1, 1, 1), (-1, -1, -1), 1.0), ((2, 1, 3), (1, -1, 0), 2.5)])
@pytest.mark.parametrize('scl', [0.0, 0.1])
def test_conv_down_up_match_torch(dev, stride, kinds, w, scl):
    from unires_amd import _ops
    torch.manual_seed(2)
    taps = [np.array(N.smooth1d(k, w * (s if k != -1 else 1) / max(stride)), dtype=np.float32)
            for k, s in zip(kinds, stride)]
    lo = (5, 6, 7)
    hi = tuple((l - 1) * s + len(t) for l, s, t in zip(lo, stride, taps))
    ker = torch.from_numpy(taps[0][:, None, None] * taps[1][None, :, None] * taps[2][None, None, :])
    ker = ker[None, None]
    dim_thick = int(np.argmax(stride))
    src = torch.rand((1, 1) + hi)
    ref = F.conv3d(src, ker, stride=stride)
    if scl:
        ref = O.apply_scaling(ref, scl, dim_thick)
    out = _ops.conv_down(src.to(dev), taps, stride, scl, dim_thick).cpu()
    assert out.shape == ref.shape and rel_err(out, ref) < TOL
    low = torch.rand((1, 1) + lo)
    ref = F.conv_transpose3d(O.apply_scaling(low, scl, dim_thick) if scl else low, ker, stride=stride)
    out = _ops.conv_up(low.to(dev), taps, stride, scl, dim_thick).cpu()
    assert out.shape == ref.shape and rel_err(out, ref) < TOL


def test_conv_dim_mismatch_is_an_error(dev):
    from unires_amd import _ops
    with pytest.raises(ValueError):
        _ops.conv_down(torch.zeros(10, 10, 10, device=dev), [np.ones(1, np.float32)] * 2 + [np.ones(3, np.float32) / 3],
                       (1, 1, 3))


@pytest.mark.parametrize('dim', [(9, 8, 7), (3, 65, 129), (1, 1, 5), (2, 1, 1)])
@pytest.mark.parametrize('vx', [(1.0, 1.0, 1.0), (0.8, 1.25, 2.0)])
def test_gradient_divergence_dtd(dev, dim, vx):
    from unires_amd import _ops, spatial
    import unires_amd as U
    torch.manual_seed(3)
    y = torch.rand(dim) * 100
    g = torch.rand((3,) + dim) * 100
    vxt = torch.tensor(vx)
    a = spatial.im_gradient(y.to(dev), vxt).cpu()
    assert rel_err(a, N.im_gradient(y, vxt)) < TOL
    a = spatial.im_divergence(g.to(dev), vxt).cpu()
    assert rel_err(a, N.im_divergence(g, vxt)) < TOL
    a = U._DtD(y.to(dev), vxt).cpu()
    assert rel_err(a, O.DtD(y, vxt)) < TOL
    a = _ops.dtd(y.to(dev), vxt, a=0.7, c=0.3).cpu()
    assert rel_err(a, 0.7 * y + 0.3 * O.DtD(y, vxt)) < TOL


def test_apply_scaling(dev):
    import unires_amd as U
    torch.manual_seed(4)
    dat = torch.rand(1, 1, 6, 7, 9)
    for dim in (0, 1, 2):
        ref = O.apply_scaling(dat, torch.tensor(0.2), dim)
        out = U._apply_scaling(dat.to(dev), torch.tensor(0.2), dim).cpu()
        assert rel_err(out, ref) < 1e-6


def test_abi_status_codes(dev, lib):
    """Errors come back as status codes + message, never as a crash."""
    import ctypes as C
    from unires_amd import _lib
    t = torch.zeros(4, 4, 4, device=dev)
    bad = _lib.i3((0, 4, 4))
    ok = _lib.i3((4, 4, 4))
    M = _lib.f12([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0])
    p = C.c_void_p(t.data_ptr())
    assert lib.unires_pull3d_affine(None, ok, M, p, ok, 0.05, None) == 1
    assert lib.unires_pull3d_affine(p, bad, M, p, ok, 0.05, None) == 2
    assert b'dimension' in lib.unires_last_error()
    assert lib.unires_dtd(p, ok, _lib.f3((1, 1, 0)), 0.0, 1.0, p, None) == 3
    h = C.c_void_p()
    r = (_lib.Repeat * 1)()
    r[0].tau = 1.0
    assert lib.unires_plan_create(C.byref(h), ok, _lib.f3((1, 1, 1)), 7, 1, r, 0.05) == 3
    assert b'Undefined method' in lib.unires_last_error()
    assert lib.unires_plan_create(C.byref(h), ok, _lib.f3((1, 1, 1)), 0, 1, r, 0.05) == 0
    assert lib.unires_proj_apply(h, 0, 9, p, p, None) == 3
    assert b'Undefined operator' in lib.unires_last_error()
    assert lib.unires_cg_solve(h, 1.0, 1.0, p, p, 5, 0.0, 1, 0, None, None, None) == 3
    assert lib.unires_plan_destroy(h) == 0


@pytest.mark.parametrize('dim', [(5, 7, 9), (6, 5, 4), (3, 4, 13), (9, 8, 70), (33, 29, 31), (40, 37, 64),
                                 (24, 50, 181), (7, 16, 16), (12, 23, 19), (4, 17, 61), (21, 64, 4)])
@pytest.mark.parametrize('shift', [0, 1, 3])
def test_identity_regime_stencil_any_shape_and_alignment(dev, dim, shift):
    """Regime A = I (unires/_project.py:76-77 + _DtD :300-317) through the flat 16-byte kernels (the x-marching one
    where a plane holds at least 64 vectors and the volume four planes, the chunked one otherwise):
    line lengths that are not multiples of 4, volumes of one chunk and of several, and p / q that
    start 4 or 12 bytes off a 16-byte boundary (the kernel aligns its vectors to q).  Also the fused
    dot product and the never-stored objective form."""
    import ctypes as C
    from unires_amd._plan import ChannelPlan
    from unires_amd._ops import _ptr, _stream
    from unires_amd._lib import check
    torch.manual_seed(sum(dim) + shift)
    n = dim[0] * dim[1] * dim[2]
    vx = (1.0, 0.8, 1.3)
    tau, rho, lam = 0.7, 1.3, 0.9
    p_cpu = torch.rand(dim)
    plan = ChannelPlan(dim, vx, [(None, tau)], 'denoising', False, device=dev)
    pbuf = torch.zeros(n + 8, device=dev)
    qbuf = torch.full((n + 8,), 7.0, device=dev)
    p = pbuf[shift:shift + n].view(dim)
    q = qbuf[(shift + 2) % 4:(shift + 2) % 4 + n].view(dim)
    p.copy_(p_cpu.to(dev))
    dot = torch.zeros((), dtype=torch.float64, device=dev)
    check(plan.lib.unires_ata_matvec(plan._h, rho, lam, _ptr(p), _ptr(q), _ptr(dot), _stream()))
    ref = tau * p_cpu + rho * lam * lam * O.DtD(p_cpu, torch.tensor(vx))
    assert rel_err(q.cpu(), ref) < 2e-6
    assert abs(float(dot) - float((p_cpu.double() * ref.double()).sum())) < 1e-5 * float(ref.abs().sum())
    # nothing written outside q
    off = (shift + 2) % 4
    assert bool((qbuf[:off] == 7.0).all()) and bool((qbuf[off + n:] == 7.0).all())


@pytest.mark.parametrize('prof', ['0', '1'])
@pytest.mark.parametrize('scl', ['0.0', '0.1'])
def test_fused_xy_conv_pass_is_bit_identical_to_the_two_passes(dev, scl, prof):
    """k_conv2d_down_xy_v4 / k_conv2d_up_xy_v4 (one kernel for the x and y passes of an isotropic
    down-sampling) form the same products in the same order as k_conv1d_*_v4 run twice: A p, At v and
    AtA p must not change by a single bit when the fused pass is switched off."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for sw in ('1', '0'):
        env = dict(os.environ, UNIRES_CONV_XY=sw)
        r = subprocess.run([sys.executable, os.path.join(here, '_conv_xy_probe.py'), scl, prof], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.split()[0] in ('A', 'At', 'AtA')])
    assert len(outs[0]) == 3 and outs[0] == outs[1]


@pytest.mark.parametrize('prof_ip', [0, 2])
@pytest.mark.parametrize('rot', [(1.25, 0.0, 0.0), (0.0, 1.3, 0.2)])
def test_isotropic_downsampling_under_large_rotations(dev, prof_ip, rot):
    """Rotations that take the grid's z axis far from the volume's (|M22| <= 0.5): the LDS-window pull
    is outside its domain, so the hybrid form (z profile inside pull / splat, x / y as 1-D passes) is
    not available and the plan is rebuilt as the separable / fused forms (build_repeat_kernels).
    Same operators as the oracle either way."""
    import unires_amd as U
    from oracle import unires_restated as O
    dim_y = (20, 18, 24)
    mat_y = torch.eye(4, dtype=torch.float64)
    mat_x = mat_y @ torch.diag(torch.tensor([2.0, 2.0, 2.0, 1.0], dtype=torch.float64))
    dim_x = tuple(d // 2 for d in dim_y)
    # rotate about the volume's centre so that most of the grid stays inside the field of view
    c = torch.eye(4, dtype=torch.float64)
    c[:3, 3] = torch.tensor([(d - 1) / 2 for d in dim_y], dtype=torch.float64)
    rigid = c @ rigid_matrix([0.3, -0.2, 0.1], list(rot)) @ torch.linalg.inv(c)
    po_o = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=prof_ip, scl=0.1)
    po_g = U._proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=prof_ip, scl=0.1, device=dev)
    torch.manual_seed(11)
    p = torch.rand(dim_y) + 0.5
    v = torch.rand(dim_x) + 0.5
    from unires_amd._plan import ChannelPlan
    plan = ChannelPlan(dim_y, (1.0, 1.0, 1.0), [(po_g, 1.0)], 'super-resolution', True, device=dev)
    for op, arg in (('A', p), ('At', v), ('AtA', p)):
        ref = O.proj_apply(op, arg[None, None], po_o, method='super-resolution')[0, 0]
        for name, out in (('plan', plan.proj_apply(0, op, arg.to(dev)).cpu()),
                          ('op-level', U._proj_apply(op, arg[None, None].to(dev), po_g, method='super-resolution')[0, 0].cpu())):
            err = float((out - ref).abs().max()) / float(ref.abs().max())
            assert err < 1e-5, '%s (%s kernels) differs from the oracle: %.3g' % (op, name, err)


def test_smaller_persistent_grids_give_the_same_bits(dev):
    """k_splat2 / k_ata1 under a grid smaller than the one the schedule was laid out with - capped by the
    runtime's occupancy answer (UNIRES_SPLAT2_RESIDENT=1), or by `unires_plan_set_concurrency` (channels of a
    y-update on streams of their own: the plan leaves its neighbours room; UNIRES_SHARE_S2 / _F1 = 1 make that cap
    16 workgroups, below the 32 this volume asks for) - walk the same tiles with the same per-tile arithmetic as
    the plain launch: At v and AtA p for thick slices along x, y and z, and the denoising regime's one-pass AtA p,
    must not change by a bit."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for env_extra in ({}, {'UNIRES_SPLAT2_RESIDENT': '1', 'UNIRES_F1_RESIDENT': '1'},
                      {'PROBE_CONCURRENCY': '3', 'UNIRES_SHARE_S2': '1', 'UNIRES_SHARE_F1': '1'}):
        r = subprocess.run([sys.executable, os.path.join(here, '_splat_wide_probe.py')], env=dict(os.environ, **env_extra),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.split()[0] in ('At', 'AtA')])
    assert len(outs[0]) == 10 and outs[0] == outs[1] == outs[2]
