# Where the host's CPU time of an ADMM iteration goes (follow-up of tools/host_time.py): main thread against the
# runtime's helper threads (process_time - thread_time), the share of the closing device synchronisation, and a
# cProfile of the main thread.
#   WL=cfg3_256c3_thick6z python tools/host_profile.py
import cProfile, io, json, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
from unires_amd._host import wait_blocking
dev = torch.device('cuda:0')


def thread_cpu():
    """{tid: (comm, cpu seconds)} of this process's threads (/proc/self/task/*/stat: utime + stime in ticks)."""
    res, hz = {}, os.sysconf('SC_CLK_TCK')
    for tid in os.listdir('/proc/self/task'):
        try:
            with open('/proc/self/task/%s/stat' % tid) as f:
                st = f.read()
            comm = st[st.index('(') + 1:st.rindex(')')]
            fields = st[st.rindex(')') + 2:].split()
            sw = 0
            with open('/proc/self/task/%s/status' % tid) as f:
                for l in f:
                    if l.startswith('voluntary_ctxt_switches'):
                        sw = int(l.split()[1])
            res[int(tid)] = (comm, (int(fields[11]) + int(fields[12])) / hz, int(fields[11]) / hz, sw)
        except (OSError, ValueError):
            pass
    return res


name = os.environ.get('WL', 'cfg3_256c3_thick6z')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[name], dev, seed=1234)
tmp = torch.zeros_like(y[0].dat)
sett.tolerance = 1e-4
sett.host_pace = int(os.environ.get('PACE', '2'))
n = int(os.environ.get('N', '40'))
out = {'workload': name, 'iterations': n, 'host_pace': sett.host_pace}
for tol in (0.0, 1e-3):
    sett.cgs_tol = tol
    obj = torch.zeros((n + 4, 3), dtype=torch.float64, device=dev)
    for yc in y:
        yc.dat.zero_()
    z.zero_(), w.zero_()
    for it in range(3):
        U._update_admm(x, y, z, w, rho, tmp, obj, it, sett)
    torch.cuda.synchronize()
    res = {}
    for closing in ('sleep', 'spin'):
        th0 = thread_cpu()
        p0, m0, t0 = time.process_time(), time.thread_time(), time.perf_counter()
        for it in range(3, 3 + n):
            U._update_admm(x, y, z, w, rho, tmp, obj, it, sett)
        p1, m1 = time.process_time(), time.thread_time()
        if closing == 'sleep':
            wait_blocking(dev)
        torch.cuda.synchronize()
        p2, m2, t2 = time.process_time(), time.thread_time(), time.perf_counter()
        res[closing] = {'wall_ms': (t2 - t0) / n * 1e3, 'process_cpu_ms': (p2 - p0) / n * 1e3,
                        'main_thread_cpu_ms': (m2 - m0) / n * 1e3,
                        'main_thread_cpu_ms_enqueue_only': (m1 - m0) / n * 1e3,
                        'other_threads_cpu_ms': ((p2 - p0) - (m2 - m0)) / n * 1e3}
        th1 = thread_cpu()
        zero = ('', 0.0, 0.0, 0)
        per = sorted(((v[1] - th0.get(t, zero)[1]) / n * 1e3, v[0], t, (v[2] - th0.get(t, zero)[2]) / n * 1e3,
                      (v[3] - th0.get(t, zero)[3]) / n) for t, v in th1.items())
        res[closing]['threads_cpu_ms'] = ['%s/%d: %.2f (user %.2f, %.0f voluntary switches per iteration)' % (comm, t, ms, us, sw)
                                          for ms, comm, t, us, sw in per[::-1] if ms > 0.02]
        res[closing]['n_threads'] = len(th1)
    out['tol=%g' % tol] = res
    if tol == 0.0:
        pr = cProfile.Profile()
        pr.enable()
        for it in range(3, 3 + n):
            U._update_admm(x, y, z, w, rho, tmp, obj, it, sett)
        pr.disable()
        wait_blocking(dev)
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
        sys.stderr.write(s.getvalue())
print(json.dumps(out))
