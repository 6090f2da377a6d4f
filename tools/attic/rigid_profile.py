# Which threads the rigid Gauss-Newton step keeps busy (follow-up of tools/r4_fit.py: 20 ms of process CPU for a
# 5 ms step) and where the main thread spends its time: per-thread CPU over 8 steps + cProfile.
#   WL=cfg3_256c3_thick6z python tools/rigid_profile.py
import cProfile, io, json, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
dev = torch.device('cuda:0')
name = os.environ.get('WL', 'cfg3_256c3_thick6z')


def thread_cpu():
    res, hz = {}, os.sysconf('SC_CLK_TCK')
    for tid in os.listdir('/proc/self/task'):
        try:
            with open('/proc/self/task/%s/stat' % tid) as f:
                st = f.read()
            fields = st[st.rindex(')') + 2:].split()
            res[int(tid)] = ((int(fields[11]) + int(fields[12])) / hz, int(fields[11]) / hz)
        except (OSError, ValueError):
            pass
    return res


x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[name], dev, seed=1234)
y = U._init_y_dat(x, y, sett)
sett.rigid_samp = 1
kw = dict(mean_correct=False, max_niter_gn=1, num_linesearch=6, samp=1)
for _ in range(2):
    U._update_rigid(x, y, sett, **kw)
torch.cuda.synchronize()
NREP = int(os.environ.get('NREP', '16'))
th0 = thread_cpu()
c0, t0 = time.process_time(), time.perf_counter()
for _ in range(NREP):
    U._update_rigid(x, y, sett, **kw)
torch.cuda.synchronize()
wall, cpu = time.perf_counter() - t0, time.process_time() - c0
th1 = thread_cpu()
main = os.getpid()
per = sorted(((v[0] - th0.get(t, (0, 0))[0]) / NREP * 1e3, t - main, (v[1] - th0.get(t, (0, 0))[1]) / NREP * 1e3)
             for t, v in th1.items())[::-1]
out = {'workload': name, 'steps': NREP, 'wall_ms': wall / NREP * 1e3, 'process_cpu_ms': cpu / NREP * 1e3,
       'threads': len(th1), 'torch_threads': torch.get_num_threads(),
       'busy_threads_ms (tid - main tid: total, user)': ['+%d: %.2f, %.2f' % (t, ms, us) for ms, t, us in per if ms > 0.05]}
print(json.dumps(out))
pr = cProfile.Profile()
pr.enable()
for _ in range(NREP):
    U._update_rigid(x, y, sett, **kw)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
sys.stderr.write(s.getvalue())
