# time the matvec of channel 0 of a workload (HIP events)
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev=torch.device('cuda:0')
wl=bench.WORKLOADS[os.environ.get('WL','cfg3_256c3_thick6z_aligned')]
x,y,z,w,rho,sett=bench.build_subject(wl,dev,seed=1234)
t=bench.time_matvec(x,y,rho,sett,reps=100)
print('matvec %.2f us  %.1f GB/s alg' % (t*1e6, bench.alg_bytes_matvec(x[0],wl['dim_y'])/t/1e9))
