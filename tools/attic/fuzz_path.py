# random problems through the fused path (matvec, RHS, y-update) against the CPU oracle
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nitorch_restated as N, unires_restated as O
from tests.helpers import SIGNED_PERMS, make_problem, oracle_structs, gpu_structs, rel_err, run_oracle_update_y, run_gpu_update_y
import unires_amd as U
dev = 'cuda:0'
g = torch.Generator().manual_seed(int(os.environ.get('SEED', '0')))
worst = 0.0
for it in range(int(os.environ.get('N', '12'))):
    dims = tuple(int(v) for v in torch.randint(10, int(os.environ.get('DMAX', '40')), (3,), generator=g))
    regime = ['sr', 'sr', 'dn', 'id'][int(torch.randint(0, 4, (1,), generator=g))]
    rot = float(torch.rand(1, generator=g)) * 0.25
    if int(torch.randint(0, 3, (1,), generator=g)) == 0:
        # translated, not rotated, line length a multiple of 4: the one-kernel matvecs (shift.hip; aligned.hip
        # when the draw is an integer shift), regime 'id': the flat stencil kernel
        rot = 0.0
        dims = (dims[0], dims[1], 4 * (dims[2] // 4 + 1))
    kw = dict(dim_y=dims, n_channels=int(torch.randint(1, 3, (1,), generator=g)), regime=regime,
              rot=rot, trans=float(torch.rand(1, generator=g)) * 4, seed=1000 + it)
    if regime == 'sr':
        kw.update(thick=int(torch.randint(2, 6, (1,), generator=g)), scl=float(torch.rand(1, generator=g)) * 0.2,
                  n_repeats=int(torch.randint(1, 3, (1,), generator=g)))
        if rot == 0.0:
            kw['thick_axes'] = [2] * kw['n_channels']
            kw['n_repeats'] = 1
        elif int(torch.randint(0, 3, (1,), generator=g)) == 0:  # profile along several axes (per-axis ratios 1..3)
            kw['iso'] = tuple(int(v) for v in torch.randint(1, 4, (3,), generator=g))
            # x-space z extent a multiple of 4 half of the time: the 16-byte 1-D passes and the fused x-y pass
            if int(torch.randint(0, 2, (1,), generator=g)) == 0:
                rz = kw['iso'][2]
                kw['dim_y'] = (dims[0], dims[1], 4 * rz * max(1, dims[2] // (4 * rz)))
        # the in-plane / through-plane slice profiles of the reference's settings (rect, triangle, Gaussian)
        kw['prof_ip'] = int(torch.randint(0, 3, (1,), generator=g))
        kw['prof_tp'] = int(torch.randint(0, 2, (1,), generator=g))
    if regime != 'id' and int(torch.randint(0, 2, (1,), generator=g)) == 0:
        # stored orientation: a random signed permutation of the voxel axes per repeat (sagittal / coronal / reflected)
        kw['orient'] = [SIGNED_PERMS[int(v)] for v in torch.randint(0, 48, (4,), generator=g)]
    try:
        prob = make_problem(**kw)
    except Exception as e:  # degenerate draw (e.g. a thick axis longer than the volume)
        print('skip', kw, type(e).__name__)
        continue
    y_ref, info_ref = run_oracle_update_y(prob, max_iter=10, tol=1e-3)
    y_gpu, info_gpu = run_gpu_update_y(prob, dev, max_iter=10, tol=1e-3)
    for c in range(len(y_ref)):
        e = rel_err(y_gpu[c].cpu(), y_ref[c])
        worst = max(worst, e)
        if e > 1e-4 or info_gpu[c][0] != info_ref[c][0]:
            print('MISMATCH', kw, c, e, info_gpu[c][0], info_ref[c][0])
print('fuzz done, worst relative error %.2e' % worst)
