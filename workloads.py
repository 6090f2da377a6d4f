"""Synthetic subjects of the benchmark configurations (BASELINE.json `configs`): the workload table and
the builder that makes one subject of a workload on the device through the HIP operator.  Shared by bench.py, the
full-size parity tests (tests/test_gpu_sizes.py, tests/test_gpu_selection.py) and the profiling scripts under tools/ -
kept out of bench.py so that an edit of the benchmark cannot change what the parity tests assert."""
import math

import torch

WORKLOADS = {
    # name: dim_y, channels, thick ratio, thick axis per channel
    'cfg3_256c3_thick6z': dict(dim_y=(256, 256, 256), C=3, thick=6, axes=(2, 2, 2)),
    'cfg3_256c3_thick6z_aligned': dict(dim_y=(256, 256, 256), C=3, thick=6, axes=(2, 2, 2), rigid='identity'),
    # the same subject translated by a fraction of a voxel per channel, no rotation (shift.hip)
    'cfg3_256c3_thick6z_shift': dict(dim_y=(256, 256, 256), C=3, thick=6, axes=(2, 2, 2), rigid='shift'),
    'cfg3_256c3_thick6xyz': dict(dim_y=(256, 256, 256), C=3, thick=6, axes=(0, 1, 2)),
    # multi-orientation thick-slice scans as files carry them (the reference's motivating case; it takes
    # mat_x as read, unires/_util.py:134-197): channel 0 axial, RAS order; channel 1 thick along world x,
    # STORED sagittally (voxel axes = world y, z, x); channel 2 thick along world y, stored coronally with
    # the first axis reversed (voxel axes = -x, z, y: LAS, det < 0).  orient[c] = (perm, flip): stored
    # voxel axis a is axis perm[a] of the axis-aligned acquisition, reversed where flip[a]
    'cfg3_256c3_thick6_orient': dict(dim_y=(256, 256, 256), C=3, thick=6, axes=(2, 0, 1),
                                     orient=(((0, 1, 2), (0, 0, 0)), ((1, 2, 0), (0, 0, 0)), ((0, 2, 1), (1, 0, 0)))),
    'cfg4_384c4_iso2': dict(dim_y=(384, 384, 384), C=4, thick=2, axes=None),
    # the same with the reference's default in-plane profile (Gaussian, struct.py:95; fan-in > 2)
    'cfg4_384c4_iso2_gauss': dict(dim_y=(384, 384, 384), C=4, thick=2, axes=None, prof_ip=2),
    'small_96c3_thick3': dict(dim_y=(96, 96, 96), C=3, thick=3, axes=(2, 2, 2)),
    # launch-bound: the device finishes every kernel before the host has enqueued the next (tools/host_time.py)
    'tiny_32c3_thick2': dict(dim_y=(32, 32, 32), C=3, thick=2, axes=(2, 2, 2)),
    # the shape of the reference's multi-channel demo (demos/demo_multi_channel.ipynb:109-113):
    # 181x217x181, three contrasts, 4 mm slices along x, y and z
    'demo_181c3_thick4xyz': dict(dim_y=(181, 217, 181), C=3, thick=4, axes=(0, 1, 2)),
    # BASELINE configs[0] / [1] shapes (BrainWeb 1 mm, 181x217x181) on the synthetic phantom:
    # single-channel denoising with A = I (R0), 3-channel 1 mm recon after coregistration (R1)
    'cfg1_181c1_denoise': dict(dim_y=(181, 217, 181), C=1, thick=1, axes=(2,), regime='id'),
    'cfg2_181c3_1mm': dict(dim_y=(181, 217, 181), C=3, thick=1, axes=(2, 2, 2), regime='dn'),
    # the pull / push operator of config 2 at the headline's size (what the single-pass kernel does at 256^3)
    'dn_256c3_1mm': dict(dim_y=(256, 256, 256), C=3, thick=1, axes=(2, 2, 2), regime='dn'),
}


def rigid_matrix(t, r):
    cx, sx, cy, sy, cz, sz = (math.cos(r[0]), math.sin(r[0]), math.cos(r[1]), math.sin(r[1]),
                              math.cos(r[2]), math.sin(r[2]))
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float64)
    M = torch.eye(4, dtype=torch.float64)
    M[:3, :3] = Rz @ Ry @ Rx
    M[:3, 3] = torch.tensor(t, dtype=torch.float64)
    return M


def orient_axes(dim, mat, perm, flip):
    """Dims and affine of the same acquisition stored with voxel axis a = old axis perm[a], reversed
    where flip[a] (mat @ Q, Q mapping stored to old voxel coordinates)."""
    Q = torch.zeros((4, 4), dtype=torch.float64)
    Q[3, 3] = 1.0
    for a in range(3):
        Q[perm[a], a] = -1.0 if flip[a] else 1.0
        if flip[a]:
            Q[perm[a], 3] = dim[perm[a]] - 1
    return tuple(int(dim[perm[a]]) for a in range(3)), mat @ Q


def phantom(dim, gen, device):
    """Sum of random ellipsoids on a zero background, built on device."""
    ax = [torch.linspace(-1, 1, d, device=device) for d in dim]
    X, Y, Z = torch.meshgrid(*ax, indexing='ij')
    vol = torch.zeros(dim, device=device)
    for _ in range(8):
        c = (torch.rand(3, generator=gen) - 0.5).tolist()
        r = (0.2 + 0.5 * torch.rand(3, generator=gen)).tolist()
        a = float(torch.rand(1, generator=gen))
        vol += a * (((X - c[0]) / r[0]) ** 2 + ((Y - c[1]) / r[1]) ** 2
                    + ((Z - c[2]) / r[2]) ** 2 < 1).float()
    return vol


def build_subject(wl, device, seed):
    """Synthetic subject: ground truth -> x = A y* + N(0, 75^2) via the HIP A
    (the demos' recipe, demos/demo_multi_channel.ipynb:173,193-202); fixed
    hyper-parameters tau = 1/75^2, lam = 4 sqrt(1/C)/mu_c, rho = sqrt(mean tau)/mean lam."""
    import unires_amd as U
    gen = torch.Generator().manual_seed(seed)
    dim_y, C, thick = wl['dim_y'], wl['C'], wl['thick']
    mat_y = torch.eye(4, dtype=torch.float64)
    if wl['axes'] is None:  # config 4: 0.5 mm recon of 1 mm isotropic inputs
        mat_y = torch.diag(torch.tensor([0.5, 0.5, 0.5, 1.0], dtype=torch.float64))
    mus = (400.0, 2000.0, 4300.0, 1000.0)
    sd = 75.0
    x, y = [], []
    for c in range(C):
        truth = phantom(dim_y, gen, device) * mus[c]
        scale = [1.0, 1.0, 1.0]
        if wl['axes'] is None:
            scale = [float(thick)] * 3
        else:
            scale[wl['axes'][c]] = float(thick)
        mat_x = mat_y @ torch.diag(torch.tensor(scale + [1.0], dtype=torch.float64))
        dim_x = tuple(int(math.floor(d / s)) for d, s in zip(dim_y, scale))
        if wl.get('orient'):
            dim_x, mat_x = orient_axes(dim_x, mat_x, *wl['orient'][c])
        u = torch.rand(6, generator=gen) * 2 - 1
        rigid = rigid_matrix((u[:3] * 5.0).tolist(), (u[3:] * 0.1).tolist())
        if wl.get('rigid') == 'identity':  # grid-aligned observations (no motion between scans)
            rigid = torch.eye(4, dtype=torch.float64)
        if wl.get('rigid') == 'shift':  # translated (+-5 mm, fractions of a voxel), not rotated
            rigid = rigid_matrix((u[:3] * 5.0).tolist(), [0.0, 0.0, 0.0])
        regime = wl.get('regime', 'sr')
        method = 'super-resolution' if regime == 'sr' else 'denoising'
        if regime == 'id':
            rigid = torch.eye(4, dtype=torch.float64)
        po = U._proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, prof_ip=wl.get('prof_ip', 0),
                          prof_tp=0, device=device)
        clean = truth if regime == 'id' else U._proj_apply('A', truth[None, None], po, method=method)[0, 0]
        noise = torch.randn(clean.shape, generator=gen).to(device) * sd
        x.append([U._input(clean + noise, mat_x, 1.0 / sd ** 2, po)])
        lam = 4.0 * math.sqrt(1.0 / C) / mus[c]
        y.append(U._output(torch.zeros(dim_y, device=device), mat_y, lam))
        del truth
    sett = U.settings()
    sett.device, sett.method, sett.do_proj = device, method, wl.get('regime', 'sr') != 'id'
    sett.cgs_max_iter, sett.cgs_tol = 20, 0.0  # fixed-iteration mode
    sett.cache_atx = False  # every step re-assembles the full RHS (no work skipped in the timed region)
    rho = float(U._step_size(x, y, sett))
    z, w = U._admm_aux(y, sett)
    return x, y, z, w, rho, sett
