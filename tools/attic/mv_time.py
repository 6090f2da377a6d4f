# time the CG matvec of a workload (HIP events): mean over channels, hot (ring 1) vs cold operands
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev=torch.device('cuda:0')
wl=bench.WORKLOADS[os.environ.get('WL','cfg3_256c3_thick6z_aligned')]
x,y,z,w,rho,sett=bench.build_subject(wl,dev,seed=1234)
for ring in (1, 2, 3):
    t=bench.time_matvec(x,y,rho,sett,reps=32,ring=ring)
    print('ring %d matvec %.2f us  %.1f GB/s alg' % (ring, t*1e6, bench.alg_bytes_matvec(x[0],wl['dim_y'])/t/1e9))
