# relative L2 / max-abs difference of the matvec and RHS of two library builds (UNIRES_LIB) on the same inputs:
#   WL=... python tools/r6_cmp.py build/ab/base_r5.so      (the other side is the in-tree library)
import os, subprocess, sys, torch
wl = os.environ.get('WL', 'cfg3_256c3_thick6z')
here = os.path.dirname(os.path.abspath(__file__))
for tag, lib in (('a', sys.argv[1]), ('b', '')):
    env = dict(os.environ, DUMP=tag, WL=wl)
    if lib:
        env['UNIRES_LIB'] = os.path.abspath(lib)
    subprocess.check_call([sys.executable, os.path.join(here, 'r6_hash.py')], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
import glob
for fa in sorted(glob.glob('/tmp/a_*.pt')):
    fb = fa.replace('/tmp/a_', '/tmp/b_')
    a, b = torch.load(fa), torch.load(fb)
    for k in ('q', 'rhs'):
        d = (a[k].double() - b[k].double())
        print('%s %s: rel L2 %.3g  max abs %.3g of max %.3g  differing voxels %d' % (os.path.basename(fa)[2:-3], k, float(d.norm() / a[k].double().norm()), float(d.abs().max()), float(a[k].abs().max()), int((d != 0).sum())))
