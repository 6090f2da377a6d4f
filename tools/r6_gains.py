# gains (nitorch get_gain, 'decreasing') of every CG iteration of the y-update's solves, in units of the tolerance -
# what the guarded stopping rule's band sees: WL=cfg3_256c3_thick6z python tools/r6_gains.py
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workloads
import unires_amd as U
dev = torch.device('cuda:0')
for name in os.environ.get('WL', 'cfg3_256c3_thick6z,demo_181c3_thick4xyz,cfg2_181c3_1mm,cfg4_384c4_iso2').split(','):
    x, y, z, w, rho, sett = workloads.build_subject(workloads.WORKLOADS[name], dev, seed=1234)
    sett.cgs_tol, sett.cgs_max_iter = 1e-3, 20
    tmp = torch.zeros_like(y[0].dat)
    for admm in range(3):
        info = []
        U._update_y(x, y, z, w, rho, tmp, sett, info=info)
        for c, (it, obj) in enumerate(info):
            g = []
            for k in range(1, len(obj)):
                rng = max(obj[:k + 1]) - min(obj[:k + 1])
                g.append(abs(obj[k - 1] - obj[k]) / rng / sett.cgs_tol)
            print('%s admm %d channel %d: %d iterations, gain / tol: %s' % (name, admm, c, it, ' '.join('%.3g' % v for v in g)))
        U._update_zw(y, z, w, rho, tmp, sett)
