# Bit-identity check across kernel changes: sha1 of the matvec output q, the 'A' output and the RHS ('At')
# of every channel of a workload on fixed seeded inputs, plus the dot.  Run before and after a kernel
# change that is meant to leave the arithmetic alone:  WL=cfg3_256c3_thick6z python tools/r6_hash.py
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unires_amd._project import _channel_plan

dev = torch.device('cuda:0')
for name in os.environ.get('WL', 'cfg3_256c3_thick6z').split(','):
    wl = bench.WORKLOADS[name]
    x, y, z, w, rho, sett = bench.build_subject(wl, dev, seed=1234)
    g = torch.Generator(device='cpu').manual_seed(7)
    out = []
    for c in range(len(x)):
        plan = _channel_plan(x[c], y[c], sett.method, sett.do_proj)
        p = torch.rand(tuple(y[c].dim), generator=g).to(dev)
        dot = torch.zeros((), dtype=torch.float64, device=dev)
        q = plan.matvec(p, rho, y[c].lam, dot=dot)
        torch.cuda.synchronize()
        h = hashlib.sha1(q.cpu().numpy().tobytes()).hexdigest()[:12]
        b = plan.rhs([xn.dat for xn in x[c]], w[c], z[c], rho, float(y[c].lam))
        torch.cuda.synchronize()
        hb = hashlib.sha1(b.cpu().numpy().tobytes()).hexdigest()[:12]
        ha = ''
        if sett.do_proj:
            a = plan.proj_apply(0, 'A', p)
            torch.cuda.synchronize()
            ha = hashlib.sha1(a.cpu().numpy().tobytes()).hexdigest()[:12]
        out.append('c%d q %s dot %.17g rhs %s A %s |q|2 %.12g |rhs|2 %.12g' % (c, h, float(dot), hb, ha, float((q.double() ** 2).sum()), float((b.double() ** 2).sum())))
        if os.environ.get('DUMP'):
            torch.save({'q': q.cpu(), 'rhs': b.cpu()}, '/tmp/%s_%s_c%d.pt' % (os.environ['DUMP'], name, c))
    print(name, ' | '.join(out))
