"""The whole ADMM iteration and both Gauss-Newton updates ABOVE toy size: the next-row
kernels (`k_jtv_scale`, `k_zw_update`, `k_masked_sse`, `k_scaling_sums`, `k_rigid_sums`, admm.hip) were only
ever compared with the oracle on volumes of ~10^4 voxels, where a tile / patch boundary bug hides easily.
Here: the geometry of bench.WORKLOADS['cfg3_256c3_thick6z'] at 96 x 90 x 102 x 3 channels (6 mm slices, even /
odd scaling on, one channel STORED in another voxel order) against the oracle - `_update_admm` (y / objective / z /
w; unires/_update.py:105-195, 396-427) for alpha = 1 and 1.5, `_update_scaling` (:270-393), `_update_rigid`
(:198-266, 541-710) - and the element-wise ADMM kernels at 256^3 x 3 against the oracle's own torch code run on
the GPU."""
import pytest
import torch

from oracle import nitorch_restated as N
from oracle import unires_restated as O
from tests.helpers import SIGNED_PERMS, gpu_structs, make_problem, oracle_structs, rel_err

pytestmark = pytest.mark.gpu

GATE = 1e-4
MID = dict(dim_y=(96, 90, 102), n_channels=3, thick=6, thick_axes=[2, 2, 2], scl=0.05, regime='sr', rot=0.1,
           trans=5.0, orient=[SIGNED_PERMS[0], SIGNED_PERMS[9], SIGNED_PERMS[0]])


def _threads():
    return torch.get_num_threads()


@pytest.mark.parametrize('alpha', [1.0, 1.5])
def test_admm_iterations_at_mid_size(dev, alpha):
    import unires_amd as U
    keep = _threads()
    torch.set_num_threads(min(16, keep))  # (the oracle's index_add_ passes anti-scale beyond ~16 threads)
    try:
        prob = make_problem(seed=41, **MID)
        xo, yo = oracle_structs(prob)
        xg, yg, sett = gpu_structs(prob, dev)
        sett.alpha = alpha
        sett.cgs_max_iter = 6
        sett.tolerance = 1e-4
        rho = torch.tensor(prob['rho'])
        zo, wo = prob['z'].clone(), prob['w'].clone()
        zg, wg = zo.clone().to(dev), wo.clone().to(dev)
        tmp = torch.zeros_like(yg[0].dat)
        n_it = 3
        obj = torch.zeros((n_it, 3), dtype=torch.float64, device=dev)
        for it in range(n_it):
            yo = O.update_y(xo, yo, zo, wo, rho, prob['method'], prob['do_proj'], cgs_max_iter=6, cgs_tol=1e-3)
            ref_obj = O.compute_nll(xo, yo, prob['method'], prob['do_proj'])
            zo, wo, tmp_o = O.update_zw(yo, zo, wo, rho, alpha=alpha)
            yg, zg, wg, tmp, obj = U._update_admm(xg, yg, zg, wg, float(rho), tmp, obj, it, sett)
            for c in range(3):
                assert rel_err(yg[c].dat.cpu(), yo[c].dat) < GATE, (it, c)
            for k in range(3):
                assert abs(obj[it, k].item() - ref_obj[k].item()) < 1e-4 * abs(ref_obj[k].item()), (it, k)
            assert rel_err(tmp.cpu(), tmp_o) < 5e-4, it
        assert rel_err(zg.cpu(), zo) < 5e-4 and rel_err(wg.cpu(), wo) < 5e-4
    finally:
        torch.set_num_threads(keep)


def test_update_scaling_at_mid_size(dev):
    import unires_amd as U
    keep = _threads()
    torch.set_num_threads(min(16, keep))
    try:
        prob = make_problem(seed=42, **MID)
        xo, yo = oracle_structs(prob)
        xg, yg, sett = gpu_structs(prob, dev)
        g = torch.Generator().manual_seed(5)
        for c in range(len(xo)):
            yo[c].dat = prob['truth'][c].clone().float() + 20
            yg[c].dat = yo[c].dat.clone().to(dev)
            po = xo[c][0].po
            keep_scl = po.scl
            po.scl = torch.tensor(0.12 if c % 2 == 0 else -0.08)  # a scaling the operator does not know yet
            dat = O.proj_apply('A', yo[c].dat[None, None], po, method=prob['method'])[0, 0]
            po.scl = keep_scl
            dat = dat + torch.randn(dat.shape, generator=g)
            dat[0, 0, :] = 0  # some masked-out voxels
            xo[c][0].dat = dat
            xg[c][0].dat = dat.clone().to(dev)
        xo, sll_o = O.update_scaling(xo, yo, method=prob['method'], max_niter_gn=1, num_linesearch=4)
        xg, sll_g = U._update_scaling(xg, yg, sett, max_niter_gn=1, num_linesearch=4)
        assert abs(sll_g.item() - sll_o.item()) < 2e-5 * abs(sll_o.item())
        for c in range(len(xo)):
            so, sg = float(xo[c][0].po.scl), float(xg[c][0].po.scl)
            assert abs(sg - so) < 1e-5, (c, so, sg)
            assert abs(so - 0.05) > 1e-3  # it moved
    finally:
        torch.set_num_threads(keep)


def test_update_rigid_at_mid_size(dev):
    import unires_amd as U
    from tests.test_gpu_path import _rigid_setup
    keep = _threads()
    torch.set_num_threads(min(16, keep))
    try:
        prob = make_problem(seed=43, **MID)
        xo, yo, xg, yg, sett, Bo = _rigid_setup(prob, dev)
        start = [[xn.po.rigid.clone() for xn in xc] for xc in xo]
        xo, sll_o = O.update_rigid(xo, yo, prob['method'], Bo, mean_correct=True, max_niter_gn=1, num_linesearch=4)
        xg, sll_g = U._update_rigid(xg, yg, sett, mean_correct=True, max_niter_gn=1, num_linesearch=4, samp=1)
        assert abs(sll_g.item() - sll_o.item()) < 1e-3 * abs(sll_o.item())
        moved = 0.0
        for c in range(len(xo)):
            Ro, Rg = xo[c][0].po.rigid, xg[c][0].po.rigid
            assert (Rg - Ro).abs().max() < 2e-3 * max(1.0, Ro[:3, 3].abs().max().item()), c
            moved = max(moved, (Ro - start[c][0]).abs().max().item())
        assert moved > 1e-3
    finally:
        torch.set_num_threads(keep)


@pytest.mark.parametrize('alpha', [1.0, 1.5])
def test_elementwise_admm_kernels_at_256(dev, alpha):
    """k_jtv_scale / k_zw_update (z, w updates and the joint-TV shrinkage image) and the prior term of the
    objective at the headline size, 256^3 x 3 channels, against the oracle's torch code executed on the GPU
    (the same arithmetic, composed of torch's element-wise kernels: no CPU pass of 200 M voxels needed)."""
    import unires_amd as U
    dim = (256, 256, 256)
    C = 3
    g = torch.Generator(device='cpu').manual_seed(7)
    mat = torch.eye(4, dtype=torch.float64)
    lam = [0.0113, 0.0021, 0.0009]
    ys = [torch.rand(dim, generator=g).to(dev) * (400.0 * (c + 1)) for c in range(C)]
    yo = [O.make_output(ys[c].clone(), mat, torch.tensor(lam[c])) for c in range(C)]
    yg = [U._output(ys[c].clone(), mat, lam[c]) for c in range(C)]
    z = torch.randn((C, 3) + dim, generator=g).to(dev) * 0.5
    w = torch.randn((C, 3) + dim, generator=g).to(dev) * 0.5
    rho = 0.37
    sett = U.settings()
    sett.device, sett.alpha = dev, alpha
    zo, wo, tmp_o = O.update_zw(yo, z.clone(), w.clone(), torch.tensor(rho), alpha=alpha)
    tmp = torch.zeros_like(ys[0])
    zg, wg, tmp = U._update_zw(yg, z.clone(), w.clone(), rho, tmp, sett)
    assert rel_err(tmp, tmp_o) < 2e-5
    assert rel_err(zg, zo) < 2e-5 and rel_err(wg, wo) < 2e-5
    del zo, wo, zg, wg, z, w
    # prior term of the objective: sum_v sqrt(sum_c lam_c^2 |D y_c|^2) in float64 (unires/_update.py:419-425)
    acc = torch.zeros(dim, dtype=torch.float32, device=dev)
    vx = N.voxel_size(mat).float()
    for c in range(C):
        acc += torch.sum((lam[c] * N.im_gradient(ys[c], vx=vx)) ** 2, dim=0)
    ref = torch.sum(torch.sqrt(acc), dtype=torch.float64).item()
    from unires_amd import _lib
    from unires_amd._lib import check, f3, i3
    from unires_amd._ops import _ptr, _stream
    from unires_amd._update import _chan_args
    lib = _lib.load()
    ptrs, lams = _chan_args(yg)
    out = torch.zeros((), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.unires_nll_prior(ptrs, lams, C, i3(dim), f3([1.0, 1.0, 1.0]), _ptr(out), _stream()))
    assert abs(out.item() - ref) < 2e-6 * abs(ref)
    # masked sum of squares (k_masked_sse): 2.75 M-voxel observation with zeros (masked out)
    x = torch.rand((256, 256, 42), generator=g).to(dev) * 100
    x[::7, :, 3] = 0
    ay = torch.rand((256, 256, 42), generator=g).to(dev) * 100
    msk = x != 0
    ref = torch.sum((x[msk] - ay[msk]) ** 2, dtype=torch.float64).item()
    sse = torch.zeros((), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.unires_masked_sse(_ptr(x), _ptr(ay), x.numel(), _ptr(sse), _stream()))
    assert abs(sse.item() - ref) < 1e-9 * abs(ref)
