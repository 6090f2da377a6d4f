# single-pass AtA kernel (ata1.hip) on a bench workload: matvec of every channel, saved or compared with a
# file written by a run of the two-kernel path (UNIRES_NO_ATA1=1), then timed (HIP events, hot / cold operands)
#   WL=cfg2_181c3_1mm SAVE=/tmp/ref.pt UNIRES_NO_ATA1=1 python tools/f1_check.py
#   WL=cfg2_181c3_1mm CMP=/tmp/ref.pt python tools/f1_check.py
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unires_amd._project import _channel_plan
dev = torch.device('cuda:0')
name = os.environ.get('WL', 'cfg2_181c3_1mm')
wl = bench.WORKLOADS[name]
x, y, z, w, rho, sett = bench.build_subject(wl, dev, seed=1234)
g = torch.Generator(device='cpu').manual_seed(7)
outs = []
for c in range(len(x)):
    pl = _channel_plan(x[c], y[c], sett.method, sett.do_proj)
    p = torch.rand(y[c].dim, generator=g).to(dev)
    q = torch.empty_like(p)
    pl.matvec(p, rho, y[c].lam, out=q)
    q2 = torch.empty_like(p)
    pl.matvec(p, rho, y[c].lam, out=q2)
    torch.cuda.synchronize()
    assert torch.equal(q, q2), 'not reproducible'
    outs.append(q.cpu())
    print('channel %d: info %s  |q| %.6e' % (c, pl.repeat_info(0) if hasattr(pl, 'repeat_info') else '', float(q.double().norm())))
if os.environ.get('SAVE'):
    torch.save(outs, os.environ['SAVE'])
if os.environ.get('CMP'):
    ref = torch.load(os.environ['CMP'])
    for c, (a, b) in enumerate(zip(outs, ref)):
        d = (a.double() - b.double())
        print('channel %d: rel L2 %.3e  max abs %.3e of %.3e' % (c, float(d.norm() / b.double().norm()), float(d.abs().max()), float(b.abs().max())))
if not os.environ.get('NOTIME'):
    for c in range(len(x)):
        t = bench.time_matvec(x, y, rho, sett, reps=32, ring=3, channels=[c])
        print('%s channel %d matvec %.2f us' % (name, c, t * 1e6))
    for ring in (1, 3):
        t = bench.time_matvec(x, y, rho, sett, reps=32, ring=ring)
        print('%s ring %d matvec %.2f us  %.1f GB/s alg  frac %.3f' % (name, ring, t * 1e6, bench.alg_bytes_matvec(x[0], wl['dim_y']) / t / 1e9, bench.alg_bytes_matvec(x[0], wl['dim_y']) / t / 8e12))
