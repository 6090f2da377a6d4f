"""CPU-only checks of the host logic and of the C-ABI surface (no compute
calls: there is no GPU in the build container)."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import rigid_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'unires_hip.h')).read()
    declared = set(re.findall(r'\b(unires_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 17
    from unires_amd import _lib
    assert declared == set(_lib.SIGNATURES), 'ctypes table and header disagree'
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = set(re.findall(r' T (unires_[a-z0-9_]+)', out))
    assert declared <= exported
    assert lib.unires_abi_version() == 1


def test_library_has_gfx950_code_object(lib):
    from unires_amd import _lib
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob


def test_missing_library_fails_loudly(monkeypatch, lib):
    from unires_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libunires_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_cpu_tensors_are_rejected(lib):
    import unires_amd as U
    from unires_amd import spatial
    with pytest.raises(RuntimeError, match='no CPU path'):
        U._DtD(torch.zeros(4, 4, 4), (1, 1, 1))
    with pytest.raises(RuntimeError):
        spatial.im_gradient(torch.zeros(4, 4, 4))


def test_undefined_operator_and_method_raise_like_the_reference(lib):
    import unires_amd as U
    eye = torch.eye(4, dtype=torch.float64)
    po = U._proj_info((8, 8, 8), eye, (8, 8, 4), eye @ torch.diag(torch.tensor([1, 1, 2, 1.], dtype=torch.float64)),
                      device='cpu')
    dat = torch.zeros(1, 1, 8, 8, 8)
    with pytest.raises(ValueError, match='Undefined operator'):
        U._proj_apply('B', dat, po)
    with pytest.raises(ValueError, match='Undefined method'):
        U._proj_apply('A', dat, po, method='sharpening')
    assert U._proj_apply('none', dat, po) is dat


@pytest.mark.parametrize('kind,w', [(-1, 1), (0, 2), (0, 3), (0, 4), (0, 6), (0, 2.5), (1, 2), (1, 3.5),
                                    (2, 2), (2, 3)])
def test_profile_closed_forms_match_oracle_quadrature(kind, w):
    from unires_amd import _kernels
    a = _kernels.smooth1d(kind, w)
    b = np.array(N.smooth1d(kind, w))
    assert a.shape == b.shape and np.abs(a - b).max() < 1e-12


def test_factorise_roundtrip_and_rejects_non_separable():
    from unires_amd import _kernels
    k = [np.array(N.smooth1d(0, 3.0)), np.array(N.smooth1d(2, 2.0)), np.array(N.smooth1d(1, 2.0))]
    dense = k[0][:, None, None] * k[1][None, :, None] * k[2][None, None, :]
    f = _kernels.factorise(dense[None, None])
    rec = f[0][:, None, None] * f[1][None, :, None] * f[2][None, None, :]
    assert np.abs(rec - dense).max() < 1e-7
    bad = dense.copy()
    bad[0, 0, 0] += 0.1
    with pytest.raises(NotImplementedError):
        _kernels.factorise(bad)


@pytest.mark.parametrize('scale,dim_x', [((1, 1, 4), (181, 217, 45)), ((4, 1, 1), (45, 217, 181)),
                                         ((1, 3, 1), (20, 7, 20)), ((2, 2, 2), (10, 10, 10))])
def test_proj_info_matches_oracle(scale, dim_x, lib):
    import unires_amd as U
    dim_y = tuple(int(d * s) if d < 100 else d for d, s in zip(dim_x, scale))
    dim_y = (181, 217, 181) if dim_x[0] in (181, 45) else dim_y
    mat_y = torch.diag(torch.tensor([0.9, 0.9, 0.9, 1.0], dtype=torch.float64))
    mat_x = mat_y @ torch.diag(torch.tensor(list(scale) + [1.0], dtype=torch.float64))
    rigid = rigid_matrix([1.0, -2.0, 0.5], [0.02, 0.03, -0.01])
    a = U._proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=2, prof_tp=0, scl=0.1,
                     device='cpu')
    b = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=2, prof_tp=0, scl=0.1)
    assert a.dim_yx == b.dim_yx and a.ratio == b.ratio and a.dim_thick == b.dim_thick
    assert a.dim_x == b.dim_x and a.dim_y == b.dim_y
    assert torch.allclose(a.mat_yx, b.mat_yx, atol=1e-12)
    assert torch.allclose(a.smo_ker.cpu(), b.smo_ker, atol=1e-7)
    from unires_amd._plan import proj_matrix
    for method in ('super-resolution', 'denoising'):
        ma, da = proj_matrix(a, method)
        mb, db = O.proj_matrix(b, method)
        assert da == tuple(db) and torch.allclose(ma, mb, atol=1e-12)


@pytest.mark.parametrize('vx_x,samp', [((0.4, 0.4, 1.0), 1), ((0.5, 0.5, 0.8), 1), ((1.0, 1.0, 3.0), 3),
                                       ((0.5, 0.5, 0.5), 1), ((0.3, 0.9, 0.6), 1)])
def test_proj_info_subsampled_matches_oracle(vx_x, samp, lib):
    """Sub-sampling branch (unires/_project.py:245-264) with the reference's default profiles
    (profile_ip = 2, profile_tp = 0, struct.py:95-96): the thick axis, the profile and the gap come
    from the ORIGINAL voxel size (:239-243 sit above the samp block), so decimation that moves the
    argmax of vx_x - (0.5, 0.5, 0.8) -> (1, 1, 0.8) - or creates a tie must not move dim_thick."""
    import unires_amd as U
    dim_y, dim_x = (40, 44, 36), (32, 30, 18)
    mat_y = torch.diag(torch.tensor([0.4, 0.4, 0.4, 1.0], dtype=torch.float64))
    mat_x = torch.diag(torch.tensor(list(vx_x) + [1.0], dtype=torch.float64))
    mat_x[:3, 3] = torch.tensor([0.3, -0.2, 0.5], dtype=torch.float64)
    rigid = rigid_matrix([0.4, -0.3, 0.2], [0.01, -0.02, 0.015])
    a = U._proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=2, prof_tp=0, scl=0.07, gap=0.1,
                     device='cpu', samp=samp)
    b = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=2, prof_tp=0, scl=0.07, gap=0.1,
                    samp=samp)
    assert a.dim_thick == b.dim_thick
    assert a.dim_x == b.dim_x and a.dim_yx == b.dim_yx and a.ratio == b.ratio
    assert torch.allclose(a.mat_x, b.mat_x, atol=1e-12) and torch.allclose(a.mat_yx, b.mat_yx, atol=1e-12)
    assert a.smo_ker.shape == b.smo_ker.shape
    assert torch.allclose(a.smo_ker.cpu(), b.smo_ker, atol=1e-7)


def test_step_size_matches_oracle(lib):
    import unires_amd as U
    x = [[U._input(None, None, 4.2e-4)], [U._input(None, None, 2.5e-4), U._input(None, None, 1.6e-4)]]
    y = [U._output(None, None, 0.0057), U._output(None, None, 0.0012)]
    s = U.settings()
    a = U._step_size(x, y, s)
    ox = [[O.make_input(torch.zeros(1, 1, 1), None, xn.tau) for xn in xc] for xc in x]
    oy = [O.make_output(torch.zeros(1, 1, 1), None, yc.lam) for yc in y]
    b = O.step_size(ox, oy)
    assert abs(float(a) - float(b)) < 1e-6 * float(b)


def test_rigid_host_algebra():
    """se(3) helpers of the rigid Gauss-Newton (host only): expm against scipy, its derivative
    against central differences, the logarithm round trip, basis shape."""
    import numpy as np
    from scipy.linalg import expm
    from unires_amd._rigid import _expm, _expm_small, _logq, affine_basis
    B = affine_basis('SE')
    assert B.shape == (6, 4, 4) and torch.all(B[:, 3, :] == 0)
    q = torch.tensor([3.0, -12.0, 0.5, 0.3, -0.25, 0.4], dtype=torch.float64)
    X = np.einsum('i,ijk->jk', q.numpy(), B.numpy())
    R, dR = _expm(q, B, grad_X=True)
    assert np.abs(expm(X) - R.numpy()).max() < 1e-12
    assert np.abs(_expm_small(np.zeros((4, 4))) - np.eye(4)).max() == 0
    assert torch.allclose(R[:3, :3] @ R[:3, :3].T, torch.eye(3, dtype=torch.float64), atol=1e-12)
    h = 1e-6
    for i in range(6):
        e = torch.zeros(6, dtype=torch.float64)
        e[i] = h
        fd = (_expm(q + e, B) - _expm(q - e, B)) / (2 * h)
        assert (fd - dR[i]).abs().max() < 1e-6
    assert torch.allclose(_logq(R, B), q, atol=1e-9)
    with pytest.raises(NotImplementedError):
        affine_basis('Aff+')


def test_fit_schedule_and_gain():
    import unires_amd as U
    s = U.settings()
    assert U._get_sched(3, s).reg_scl.tolist() == [32.0, 16.0, 8.0, 4.0]
    assert U._get_sched(1, U.settings()).reg_scl.tolist() == [4.0]
    from unires_amd.optim import get_gain
    assert abs(float(get_gain([5.481, 4.983, 4.706], 'decreasing')) - 0.3574) < 1e-3  # demo trace
    assert float(get_gain([1.0])) == float('inf')


def test_device_guard_finds_the_tensor_device_and_is_transparent_on_cpu():
    """Every library entry point runs under _ops.on_device: the device of its tensors becomes the
    current HIP device for the call (plan workspace, streams).  Without a CUDA tensor in the
    arguments the wrapper must be a plain call."""
    import torch
    from types import SimpleNamespace
    from unires_amd import _ops, _update, _plan
    calls = []

    @_ops.on_device
    def f(a, b=None):
        calls.append((a, b))
        return 7

    assert f(torch.zeros(2), b=[SimpleNamespace(dat=torch.zeros(1))]) == 7 and len(calls) == 1
    assert _ops._find_device(torch.zeros(3)) is None
    assert _ops._find_device([[SimpleNamespace(dat=torch.zeros(1))]]) is None
    fake = SimpleNamespace(device=torch.device('cuda', 3))
    assert _ops._find_device([fake]) == torch.device('cuda', 3)
    # the decorated entry points keep their names / docstrings
    assert _update._update_y.__name__ == '_update_y' and 'UPDATE: y' in _update._update_y.__doc__
    assert _plan.ChannelPlan.matvec.__name__ == 'matvec'


def test_spatial_facade_recovers_the_affine_of_a_dense_grid():
    """nitorch-style calls hand over a dense (1, X, Y, Z, 3) grid; the path only ever builds affine
    grids (unires/_project.py:159), so the facade recovers the matrix and refuses anything else."""
    import torch
    from unires_amd import spatial
    from tests.helpers import rigid_matrix
    M = rigid_matrix([1.5, -2.0, 0.7], [0.03, -0.05, 0.08])
    M[:3, :3] *= 1.3
    shape = (7, 6, 5)
    g = spatial.affine_grid(M, shape)
    assert g.shape == shape + (3,) and g.dtype == torch.float32
    mat, shp = spatial._affine_of_grid(g[None])
    assert shp == shape and torch.allclose(mat, M, atol=1e-5)
    bent = g.clone()
    bent[3, 2, 2, 0] += 0.5
    import pytest
    with pytest.raises(NotImplementedError):
        spatial._affine_of_grid(bent)
    # a deformation away from the corners / centre (where r2's four probes sat) is caught as well
    bent2 = g.clone()
    bent2[1, 1, 1, 2] -= 0.3
    with pytest.raises(NotImplementedError):
        spatial._affine_of_grid(bent2)
    from unires_amd._util import _bids_name
    assert _bids_name('/a/b/sub-01_T1w.nii') == '/a/b/sub-01_space-unires_T1w.nii'
    assert _bids_name('img.nii.gz') == 'space-unires_img.nii.gz'


def test_orientation_relabelling_of_every_signed_permutation(lib):
    """unires_orient_of (host arithmetic of the plan's canonicalisation, csrc/orient.hip): for an observation
    stored with its voxel axes permuted / reversed (tests.helpers.orient_axes: what a sagittal / coronal / LAS
    file hands the reference, unires/_util.py:134-197) composed with a small rigid, the relabelling returned
    brings the linear part of M = mat_y \\ rigid mat_yx back to a positive dominant diagonal - and it is the
    identity for axis-aligned, right-handed storage, so that those operators are used exactly as given."""
    from tests.helpers import SIGNED_PERMS, orient_axes
    from unires_amd import _lib
    dim_y = (16, 14, 12)
    mat_y = torch.eye(4, dtype=torch.float64)
    rigid = rigid_matrix([0.7, -0.4, 0.3], [0.06, -0.05, 0.08])
    for thick_axis in range(3):
        sc = [1.0, 1.0, 1.0, 1.0]
        sc[thick_axis] = 3.0
        mat_x0 = mat_y @ torch.diag(torch.tensor(sc, dtype=torch.float64))
        dim_x0 = tuple(int(d // s) for d, s in zip(dim_y, sc))
        for perm, flip in SIGNED_PERMS:
            dim_x, mat_x = orient_axes(dim_x0, mat_x0, perm, flip)
            po = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid)
            M, _ = O.proj_matrix(po, 'super-resolution')
            m12 = _lib.c_f32x12(*[float(v) for v in M[:3, :].reshape(-1).tolist()])
            p_out, f_out = _lib.c_i32x3(), _lib.c_i32x3()
            assert lib.unires_orient_of(m12, p_out, f_out) == 0
            p_out, f_out = list(p_out), list(f_out)
            assert sorted(p_out) == [0, 1, 2]
            lin = M[:3, :3]
            canon = torch.stack([lin[:, p_out[j]] * (-1.0 if f_out[j] else 1.0) for j in range(3)], dim=1)
            for j in range(3):
                col = canon[:, j]
                assert col[j] > 0 and abs(col[j]) == col.abs().max(), (perm, flip, p_out, f_out)
            if perm == (0, 1, 2) and flip == (0, 0, 0):
                assert p_out == [0, 1, 2] and f_out == [0, 0, 0]
            else:  # stored axis a is acquisition axis perm[a]: canonical axis perm[a] is the caller's axis a
                assert all(p_out[perm[a]] == a and f_out[perm[a]] == flip[a] for a in range(3))
    bad = _lib.c_f32x12(*([float('nan')] * 12))
    assert lib.unires_orient_of(bad, _lib.c_i32x3(), _lib.c_i32x3()) != 0


def test_host_shares_and_pacer_without_a_gpu():
    """_host.py: every rank gets a contiguous, disjoint share of the cores (at least one), the thread cap
    is applied once, and the pacer / blocking wait are no-ops where there is no device."""
    import torch
    from unires_amd import _host
    cores = list(range(3, 3 + 64))
    shares = [_host.core_share(cores, 8, r) for r in range(8)]
    assert all(len(s) == 8 for s in shares)
    flat = [c for s in shares for c in s]
    assert sorted(flat) == cores and len(set(flat)) == 64
    assert _host.core_share(list(range(5)), 8, 7) and len(_host.core_share(list(range(5)), 8, 7)) == 1
    assert _host._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    keep = torch.get_num_threads()
    try:
        cfg = _host.configure_host(local_rank=0, world=1)
        # one rank: no pinning, the process keeps its thread pool, no sticky caps in host sections
        assert cfg['threads'] == keep and cfg['cpus'] is None and torch.get_num_threads() == keep
        assert _host.batch_mode == (os.environ.get('UNIRES_LIGHT_HOST', '0') == '1')
    finally:
        torch.set_num_threads(keep)
    if not torch.cuda.is_available():
        p = _host.Pacer(2)
        for _ in range(5):
            p.step()
        p.drain()
        _host.wait_blocking()


def test_objective_trace_from_the_ring():
    """_plan._trace: all values while they fit the library's ring, else the last len(ring), oldest first
    (budgets beyond 4 096 iterations)."""
    from unires_amd._plan import _trace
    ring = list(range(5))
    assert _trace(ring, 2) == [0, 1, 2] and _trace(ring, 4) == [0, 1, 2, 3, 4]
    assert _trace([5, 6, 2, 3, 4], 6) == [2, 3, 4, 5, 6]  # iteration k lives in slot k mod 5
    assert _trace([10, 6, 7, 8, 9], 10) == [6, 7, 8, 9, 10]


def test_blas_pools_are_capped_once_and_for_good():
    """`cap_blas` (the Gauss-Newton steps' 4 x 4 / 6 x 6 algebra must not wake OpenBLAS worker pools that then
    spin: 20 -> 5 ms of CPU per rigid step, profiles/r05_rigid_profile.txt): in batch mode (several ranks on the
    host, or UNIRES_LIGHT_HOST=1) after a `light_host` section every loaded BLAS - numpy's and scipy's copy - runs
    on one thread and stays there; UNIRES_BLAS_THREADS=0 leaves the pools alone - and so does a lone process
    (no process-wide side effect of a library call outside batch mode)."""
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from threadpoolctl import threadpool_info\n"
        "before = [i['num_threads'] for i in threadpool_info() if i['user_api'] == 'blas']\n"
        "from unires_amd import _host\n"
        "_host.batch_mode = _host.batch_mode or %r\n"
        "f = _host.light_host(lambda: np.linalg.solve(np.eye(6), np.ones(6)).sum())\n"
        "assert f() == 6.0\n"
        "n1 = [i['num_threads'] for i in threadpool_info() if i['user_api'] == 'blas']\n"
        "f()\n"
        "n2 = [i['num_threads'] for i in threadpool_info() if i['user_api'] == 'blas']\n"
        "print(before, n1, n2)\n")
    lone, code = code % (ROOT, False), code % (ROOT, True)
    r = subprocess.run([sys.executable, '-c', lone], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    before, n1, n2 = eval('(' + r.stdout.strip().replace('] [', '], [') + ')')
    assert n1[:len(before)] == before and n2 == n1, (before, n1, n2)  # a lone process: pools untouched
    r = subprocess.run([sys.executable, '-c', lone], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, UNIRES_LIGHT_HOST='1'))
    assert r.returncode == 0, r.stderr[-2000:]
    before, n1, n2 = eval('(' + r.stdout.strip().replace('] [', '], [') + ')')
    assert n1 and n1 == n2 and all(v == 1 for v in n1), (before, n1, n2)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    before, n1, n2 = eval('(' + r.stdout.strip().replace('] [', '], [') + ')')
    assert n1 and n1 == n2 and all(v == 1 for v in n1), (before, n1, n2)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, UNIRES_BLAS_THREADS='0'))
    assert r.returncode == 0, r.stderr[-2000:]
    before, n1, n2 = eval('(' + r.stdout.strip().replace('] [', '], [') + ')')
    assert n1[:len(before)] == before and n2 == n1, (before, n1, n2)


def test_import_sets_the_runtimes_signal_pool_unless_told_otherwise():
    """`_host.tune_runtime`: ROC_SIGNAL_POOL_SIZE=4096 from the package import, a value the user set
    is kept, UNIRES_NO_RUNTIME_TUNING=1 leaves the environment alone."""
    import sys
    code = "import sys, os; sys.path.insert(0, %r); import unires_amd; print(os.environ.get('ROC_SIGNAL_POOL_SIZE'))" % ROOT
    base = {k: v for k, v in os.environ.items() if k not in ('ROC_SIGNAL_POOL_SIZE', 'UNIRES_NO_RUNTIME_TUNING')}
    for extra, want in (({}, '4096'), ({'ROC_SIGNAL_POOL_SIZE': '128'}, '128'),
                        ({'UNIRES_NO_RUNTIME_TUNING': '1'}, 'None')):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300,
                           env=dict(base, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.strip().splitlines()[-1] == want, (extra, r.stdout)


def test_import_sets_the_signal_pool_only_when_it_can():
    """`import unires_amd` exports ROC_SIGNAL_POOL_SIZE=4096 unless the user set it (or opted out); the variable is
    read once, when the HIP runtime starts - so an import that comes AFTER the first CUDA/HIP call (an embedding
    application) warns once, with the measured cost, instead of silently doing nothing
    (VERDICT r5 weak 9).  `torch.cuda.is_initialized()` stands in for "the runtime is up" here (no GPU needed)."""
    import sys
    code = (
        "import sys, os, warnings; sys.path.insert(0, %r)\n"
        "import torch\n"
        "if %r: torch.cuda.is_initialized = lambda: True\n"
        "with warnings.catch_warnings(record=True) as w:\n"
        "    warnings.simplefilter('always')\n"
        "    import unires_amd\n"
        "print(os.environ.get('ROC_SIGNAL_POOL_SIZE'), sum('ROC_SIGNAL_POOL_SIZE' in str(x.message) for x in w))\n")
    env = {k: v for k, v in os.environ.items() if k not in ('ROC_SIGNAL_POOL_SIZE', 'UNIRES_NO_RUNTIME_TUNING')}

    def run(late, **extra):
        r = subprocess.run([sys.executable, '-c', code % (ROOT, late)], capture_output=True, text=True, timeout=300,
                           env=dict(env, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.split()
    assert run(False) == ['4096', '0']                                 # first: set, silent
    assert run(True) == ['4096', '1']                                  # late: one warning
    assert run(True, ROC_SIGNAL_POOL_SIZE='256') == ['256', '0']       # the user's value wins, nothing to say
    assert run(True, UNIRES_NO_RUNTIME_TUNING='1') == ['None', '0']    # opted out
