# which threads of the process burn CPU during one rigid Gauss-Newton step (per-thread utime + stime from /proc)
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[os.environ.get('WL', 'cfg3_256c3_thick6z')], dev, seed=1234)
y = U._init_y_dat(x, y, sett)
sett.scaling, sett.unified_rigid, sett.rigid_samp = True, True, 1
def threads_cpu():
    out = {}
    for t in os.listdir('/proc/self/task'):
        try:
            st = open('/proc/self/task/%s/stat' % t).read()
            name = st[st.index('(') + 1:st.rindex(')')]
            f = st.rsplit(')', 1)[1].split()
            out[t] = (name, (int(f[11]) + int(f[12])) / os.sysconf('SC_CLK_TCK'))
        except Exception:
            pass
    return out
for rep in range(4):
    torch.cuda.synchronize()
    a, c0, t0 = threads_cpu(), time.process_time(), time.perf_counter()
    U._update_rigid(x, y, sett, mean_correct=False, max_niter_gn=1, num_linesearch=6, samp=1)
    torch.cuda.synchronize()
    wall, cpu, b = time.perf_counter() - t0, time.process_time() - c0, threads_cpu()
    busy = sorted(((b[k][1] - a.get(k, ('', 0.0))[1], b[k][0]) for k in b), reverse=True)[:6]
    print(json.dumps({'rep': rep, 'wall_ms': wall * 1e3, 'cpu_ms': cpu * 1e3, 'threads': len(b), 'busiest': [(round(v * 1e3), n) for v, n in busy]}))
