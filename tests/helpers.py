"""Shared test scaffolding: seeded synthetic y-update problems, built once and
fed to both the CPU oracle (oracle/) and the HIP path (unires_amd/)."""
import math

import torch

from oracle import unires_restated as O
from oracle.nitorch_restated import affine_grid as N_affine_grid


def rigid_matrix(t, r):
    """4x4 float64 rigid: translation t (mm), rotations r (rad) about x, y, z."""
    cx, sx = math.cos(r[0]), math.sin(r[0])
    cy, sy = math.cos(r[1]), math.sin(r[1])
    cz, sz = math.cos(r[2]), math.sin(r[2])
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float64)
    M = torch.eye(4, dtype=torch.float64)
    M[:3, :3] = Rz @ Ry @ Rx
    M[:3, 3] = torch.tensor(t, dtype=torch.float64)
    return M


def orient_axes(dim, mat, perm, flip):
    """The same acquisition STORED with its voxel axes permuted / reversed (sagittal or coronal
    storage, LAS vs RAS): new voxel axis a is old axis perm[a], reversed where flip[a].  Returns the
    stored dims and affine (mat @ Q, Q mapping stored to old voxel coordinates) - what
    ``_read_image`` hands the reference as it is (unires/_util.py:134-197)."""
    Q = torch.zeros((4, 4), dtype=torch.float64)
    Q[3, 3] = 1.0
    for a in range(3):
        Q[perm[a], a] = -1.0 if flip[a] else 1.0
        if flip[a]:
            Q[perm[a], 3] = dim[perm[a]] - 1
    return tuple(int(dim[perm[a]]) for a in range(3)), mat @ Q


SIGNED_PERMS = [(p, f) for p in ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))
                for f in ((0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 0, 1), (1, 1, 0), (1, 1, 1))]


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def smooth_volume(dim, gen, scale):
    """Smooth random phantom: low-pass filtered noise, non-negative."""
    v = torch.rand(dim, generator=gen)
    k = torch.ones(1, 1, 3, 3, 3) / 27.0
    v = torch.nn.functional.conv3d(v[None, None], k, padding=1)[0, 0]
    return (v * scale).float()


def fov_margin(po, method):
    """Smallest distance of any grid coordinate to the +-5e-2 in-FOV thresholds.
    The reference's mask is discontinuous there: a coordinate that ties with a
    threshold in float32 flips with the last-ulp rounding of the coordinate
    arithmetic (torch-CPU matmul vs cuBLAS vs FMA), so parity problems are drawn
    away from such ties."""
    mat, dim = O.proj_matrix(po, method)
    g = N_affine_grid(mat.float(), dim)
    m = float('inf')
    for d, n in enumerate(po.dim_y):
        for thr in (-5e-2, n - 1 + 5e-2):
            m = min(m, (g[..., d] - thr).abs().min().item())
    return m


def make_problem(dim_y=(16, 14, 12), n_channels=1, thick=3, seed=0, scl=0.0, regime='sr',
                 n_repeats=1, vx_y=1.0, rot=0.05, trans=0.7, noise_sd=20.0, prof_tp=0,
                 prof_ip=0, thick_axes=None, aniso=None, iso=False, orient=None):
    """A synthetic multi-channel y-update problem.

    regime 'sr': thick-slice observations (ratio ``thick`` along a per-channel axis),
    'dn': same-resolution observations with a rigid misalignment (pull/push only),
    'id': do_proj False (A = I).
    ``orient``: per (channel, repeat) index c * n_repeats + n (cycled), a (perm, flip) pair: the
    observation is stored with its voxel axes permuted / reversed (``orient_axes``).
    """
    def stored(c, n, dim, mat):
        if orient is None:
            return dim, mat
        perm, flip = orient[(c * n_repeats + n) % len(orient)]
        return orient_axes(dim, mat, perm, flip)

    gen = torch.Generator().manual_seed(seed)
    mat_y = torch.diag(torch.tensor([vx_y, vx_y, vx_y, 1.0], dtype=torch.float64))
    if aniso is not None:
        mat_y = torch.diag(torch.tensor(list(aniso) + [1.0], dtype=torch.float64))
    C = n_channels
    truth = [smooth_volume(dim_y, gen, 400.0 * (c + 1)) for c in range(C)]
    chans = []
    for c in range(C):
        reps = []
        for n in range(n_repeats):
            for _attempt in range(20):
                u = torch.rand(6, generator=gen) * 2 - 1
                rigid = rigid_matrix((u[:3] * trans).tolist(), (u[3:] * rot).tolist())
                if regime == 'id':
                    break
                if regime == 'sr':
                    ax_ = (thick_axes[c] if thick_axes is not None else (2 - c - n) % 3)
                    sc_ = [1.0, 1.0, 1.0]
                    sc_[ax_] = float(thick)
                    if iso:
                        sc_ = [float(v) for v in iso] if isinstance(iso, (tuple, list)) else [float(thick)] * 3
                    mx_ = mat_y @ torch.diag(torch.tensor(sc_ + [1.0], dtype=torch.float64))
                    dx_ = tuple(int(math.floor(d / s_)) for d, s_ in zip(dim_y, sc_))
                    dx_, mx_ = stored(c, n, dx_, mx_)
                    po_ = O.proj_info(dim_y, mat_y, dx_, mx_, rigid=rigid, prof_ip=prof_ip,
                                      prof_tp=prof_tp)
                    if fov_margin(po_, 'super-resolution') > 1e-4:
                        break
                else:
                    dx_, mx_ = stored(c, n, tuple(dim_y), mat_y)
                    po_ = O.proj_info(dim_y, mat_y, dx_, mx_, rigid=rigid)
                    if fov_margin(po_, 'denoising') > 1e-4:
                        break
            if regime == 'sr':
                ax = (thick_axes[c] if thick_axes is not None else (2 - c - n) % 3)
                scale = [1.0, 1.0, 1.0]
                scale[ax] = float(thick)
                if iso:  # BASELINE config 4: every axis coarser (ratio thick, thick, thick, or as given)
                    scale = [float(v) for v in iso] if isinstance(iso, (tuple, list)) else [float(thick)] * 3
                D = torch.diag(torch.tensor(scale + [1.0], dtype=torch.float64))
                mat_x = mat_y @ D
                dim_x = tuple(int(math.floor(d / s)) for d, s in zip(dim_y, scale))
                dim_x, mat_x = stored(c, n, dim_x, mat_x)
                po = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=prof_ip,
                                 prof_tp=prof_tp, scl=scl)
                method = 'super-resolution'
            elif regime == 'dn':
                dim_x, mat_x = stored(c, n, tuple(dim_y), mat_y.clone())
                po = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid)
                method = 'denoising'
            else:
                mat_x, dim_x, rigid = mat_y.clone(), tuple(dim_y), torch.eye(4, dtype=torch.float64)
                po = O.proj_info(dim_y, mat_y, dim_x, mat_x)
                method = 'denoising'
            do_proj = regime != 'id'
            clean = O.proj_apply('A' if do_proj else 'none', truth[c][None, None], po,
                                 method=method)[0, 0]
            dat = clean + noise_sd * torch.randn(clean.shape, generator=gen)
            reps.append(dict(dim_x=dim_x, mat_x=mat_x, rigid=rigid, scl=scl if regime == 'sr' else 0.0,
                             tau=1.0 / (noise_sd * (1.0 + 0.3 * n)) ** 2, dat=dat.float(),
                             prof_ip=prof_ip, prof_tp=prof_tp))
        mu = float(truth[c].mean())
        chans.append(dict(reps=reps, lam=4.0 * math.sqrt(1.0 / C) / mu))
    all_tau = torch.tensor([r['tau'] for ch in chans for r in ch['reps']], dtype=torch.float32)
    all_lam = torch.tensor([ch['lam'] for ch in chans], dtype=torch.float32)
    rho = float(torch.sqrt(all_tau.mean()) / all_lam.mean())
    y0 = [(truth[c] * 0.8 + 20.0 * torch.rand(dim_y, generator=gen)).float() for c in range(C)]
    z = 0.05 * torch.randn((C, 3) + tuple(dim_y), generator=gen)
    w = 0.05 * torch.randn((C, 3) + tuple(dim_y), generator=gen)
    return dict(dim_y=tuple(dim_y), mat_y=mat_y, chans=chans, rho=rho, y0=y0, z=z.float(),
                w=w.float(), method=method, do_proj=do_proj, truth=truth)


# ---------------------------------------------------------------------------
# oracle side
# ---------------------------------------------------------------------------
def oracle_structs(prob):
    x, y = [], []
    for c, ch in enumerate(prob['chans']):
        xc = []
        for r in ch['reps']:
            po = O.proj_info(prob['dim_y'], prob['mat_y'], r['dim_x'], r['mat_x'], rigid=r['rigid'],
                             prof_ip=r['prof_ip'], prof_tp=r['prof_tp'], scl=r['scl'])
            xc.append(O.make_input(r['dat'].clone(), r['mat_x'], torch.tensor(r['tau']), po))
        x.append(xc)
        y.append(O.make_output(prob['y0'][c].clone(), prob['mat_y'], torch.tensor(ch['lam'])))
    return x, y


def run_oracle_update_y(prob, max_iter=20, tol=1e-3, jacobi=False, fft=False):
    x, y = oracle_structs(prob)
    rho = torch.tensor(prob['rho'])
    y, info = O.update_y(x, y, prob['z'].clone(), prob['w'].clone(), rho, prob['method'],
                         prob['do_proj'], cgs_max_iter=max_iter, cgs_tol=tol, return_info=True,
                         jacobi=jacobi, fft=fft)
    return [yc.dat for yc in y], info


# ---------------------------------------------------------------------------
# HIP side
# ---------------------------------------------------------------------------
def gpu_structs(prob, device):
    import unires_amd as U
    x, y = [], []
    for c, ch in enumerate(prob['chans']):
        xc = []
        for r in ch['reps']:
            po = U._proj_info(prob['dim_y'], prob['mat_y'], r['dim_x'], r['mat_x'],
                              rigid=r['rigid'], prof_ip=r['prof_ip'], prof_tp=r['prof_tp'],
                              scl=r['scl'], device=device)
            xc.append(U._input(r['dat'].to(device), r['mat_x'], r['tau'], po))
        x.append(xc)
        y.append(U._output(prob['y0'][c].clone().to(device), prob['mat_y'], ch['lam']))
    sett = U.settings()
    sett.device = device
    sett.method = prob['method']
    sett.do_proj = prob['do_proj']
    return x, y, sett


def run_gpu_update_y(prob, device='cuda:0', max_iter=20, tol=1e-3, stop='max_gain', precond='none'):
    import unires_amd as U
    x, y, sett = gpu_structs(prob, device)
    sett.cgs_max_iter, sett.cgs_tol, sett.cgs_stop = max_iter, tol, stop
    sett.cgs_precond = precond
    z, w = prob['z'].to(device), prob['w'].to(device)
    tmp = torch.zeros_like(y[0].dat)
    info = []
    U._update_y(x, y, z, w, prob['rho'], tmp, sett, info=info)
    torch.cuda.synchronize()
    return [yc.dat for yc in y], info


# ---- full-size parity helpers (one channel of a bench workload as oracle structs; moved here from bench.py in
# round 6 so that the benchmark and the parity assertions do not share editable code) ----
def oracle_channel(wl, dim_y, seed=0, channel=None):
    """One channel of workload ``wl`` at size ``dim_y`` as oracle structs (+ the raw pieces).
    ``channel`` picks that channel's thick axis (default: z, the headline configuration's)."""
    from oracle import unires_restated as O
    thick = wl['thick']
    gen = torch.Generator().manual_seed(seed)
    mat_y = torch.eye(4, dtype=torch.float64)
    scale = [1.0, 1.0, 1.0]
    if wl['axes'] is None:
        scale = [float(thick)] * 3
    else:
        scale[2 if channel is None else wl['axes'][channel]] = float(thick)
    if wl['axes'] is None:
        mat_y = torch.diag(torch.tensor([0.5, 0.5, 0.5, 1.0], dtype=torch.float64))
    mat_x = mat_y @ torch.diag(torch.tensor(scale + [1.0], dtype=torch.float64))
    dim_x = tuple(int(math.floor(d / s)) for d, s in zip(dim_y, scale))
    if wl.get('orient') and channel is not None:  # the channel's stored voxel order (sagittal / coronal / reflected)
        dim_x, mat_x = orient_axes(dim_x, mat_x, *wl['orient'][channel])
    u = torch.rand(6, generator=gen) * 2 - 1
    rigid = rigid_matrix((u[:3] * 5.0).tolist(), (u[3:] * 0.1).tolist())
    po = O.proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=wl.get('prof_ip', 0), prof_tp=0)
    dat_x = torch.rand(dim_x, generator=gen) * 400
    tau, lam = 1 / 75.0 ** 2, 4.0 * math.sqrt(1 / 3.0) / 400.0
    xc = [O.make_input(dat_x, mat_x, torch.tensor(tau), po)]
    yc = O.make_output(torch.zeros(dim_y), mat_y, torch.tensor(lam))
    b = torch.rand(dim_y, generator=gen)
    return dict(mat_y=mat_y, mat_x=mat_x, dim_x=dim_x, rigid=rigid, xc=xc, yc=yc, b=b, tau=tau, lam=lam,
                dat_x=dat_x, po=po)


def oracle_lhs(wl, P, rho=0.9):
    from oracle import nitorch_restated as N
    from oracle import unires_restated as O
    regime = wl.get('regime', 'sr')
    method = 'super-resolution' if regime == 'sr' else 'denoising'
    vx = N.voxel_size(P['mat_y']).float()
    return lambda d: O.proj('AtA', d, P['xc'], P['yc'], method=method, do=regime != 'id',
                            rho=torch.tensor(rho), vx_y=vx)


def fov_tie_voxels(wl, P, eps=1e-4, reach=2):
    """Output voxels that a grid point within ``eps`` of an in-FOV threshold can reach.  The
    reference's mask is discontinuous there: which side a float32 coordinate falls on depends on
    the last-ulp rounding of the coordinate arithmetic (torch-CPU matmul vs FMA chain), so the
    matvec legitimately differs by one grid point's worth in these voxels."""
    from oracle import nitorch_restated as N
    from oracle import unires_restated as O
    regime = wl.get('regime', 'sr')
    method = 'super-resolution' if regime == 'sr' else 'denoising'
    mat, dim = O.proj_matrix(P['po'], method)
    g = N.affine_grid(mat.float(), dim)
    dim_y = tuple(P['b'].shape)
    near = torch.zeros(g.shape[:3], dtype=torch.bool)
    for d, n in enumerate(dim_y):
        for thr in (-5e-2, n - 1 + 5e-2):
            near |= (g[..., d] - thr).abs() < eps
    pts = g[near]
    bad = torch.zeros(dim_y, dtype=torch.bool)
    for pt in pts:
        lo = [int(max(0, math.floor(float(v)) - reach + 1)) for v in pt]
        hi = [int(min(n, math.floor(float(v)) + reach + 1)) for v, n in zip(pt, dim_y)]
        if all(h > l for l, h in zip(lo, hi)):
            bad[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
    return bad, int(near.sum())


def matvec_parity(wl, P, q_cpu, device, rho=0.9):
    """float32 agreement of the HIP matvec with the oracle on the same operator and input
    (relative L2 error, gate 1e-4, and max-abs error)."""
    import unires_amd as U
    from unires_amd._project import _channel_plan
    regime = wl.get('regime', 'sr')
    method = 'super-resolution' if regime == 'sr' else 'denoising'
    dim_y = tuple(P['b'].shape)
    po_g = U._proj_info(dim_y, P['mat_y'], P['dim_x'], P['mat_x'], rigid=P['rigid'],
                        prof_ip=wl.get('prof_ip', 0), prof_tp=0, device=device)
    xg = [U._input(P['dat_x'].to(device), P['mat_x'], P['tau'], po_g)]
    yg = U._output(torch.zeros(dim_y, device=device), P['mat_y'], P['lam'])
    plan = _channel_plan(xg, yg, method, regime != 'id')
    q_gpu = plan.matvec(P['b'].to(device), rho, P['lam']).cpu()
    diff = (q_gpu.double() - q_cpu.double())
    ties, n_near = fov_tie_voxels(wl, P)
    keep = ~ties
    return {'rel_err': float(diff.norm() / q_cpu.double().norm()),
            'max_abs': float(diff.abs().max()), 'ref_max_abs': float(q_cpu.abs().max()),
            'rel_err_away_from_fov_ties': float(diff[keep].norm() / q_cpu.double()[keep].norm()),
            'max_abs_away_from_fov_ties': float(diff[keep].abs().max()),
            'fov_tie_grid_points': n_near, 'fov_tie_voxels_excluded': int(ties.sum()),
            'what': 'HIP ata_matvec vs oracle _proj(AtA) on the same %dx%dx%d operator and input; '
                    '"away from ties" leaves out the output voxels within reach of grid points whose '
                    'coordinate lies within 1e-4 of an in-FOV threshold (the reference mask is '
                    'discontinuous there)' % dim_y}
