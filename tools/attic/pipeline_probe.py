# Feasibility probe: the CG iterations of several channels as a two-lane pipeline - all matvecs on one stream (M), all
# vector updates on another (V), channel c's vector work running under channel c+1's matvec - against the serial
# order.  The matvec is the library's (unires_ata_matvec); the vector updates are stand-ins made of torch kernels with
# the same traffic (r -= a q: 3 passes; x += a p; p = r + b p: 5 passes).  Says what the hardware's co-scheduling is
# worth before the library's CG driver is touched.
#   WL=cfg3_256c3_thick6z ITERS=20 python tools/pipeline_probe.py
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
from unires_amd._project import _channel_plan
dev = torch.device('cuda:0')
name = os.environ.get('WL', 'cfg3_256c3_thick6z')
iters = int(os.environ.get('ITERS', '20'))
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[name], dev, seed=1234)
C = len(x)
plans = [_channel_plan(x[c], y[c], sett.method, sett.do_proj) for c in range(C)]
lams = [float(y[c].lam) for c in range(C)]
g = torch.Generator(device='cpu').manual_seed(1)
P = [torch.rand(y[c].dat.shape, generator=g).to(dev) for c in range(C)]
Q = [torch.empty_like(P[c]) for c in range(C)]
R = [torch.rand(y[c].dat.shape, generator=g).to(dev) for c in range(C)]
X = [torch.zeros_like(P[c]) for c in range(C)]


def vec(c):
    R[c].add_(Q[c], alpha=-1e-3)          # r -= a q              (3 passes)
    X[c].add_(P[c], alpha=1e-3)           # x += a p              (3 passes)
    P[c].mul_(0.5).add_(R[c])             # p = r + b p           (2 + 3 passes; the library's fused form does 8 in all)


def serial():
    for k in range(iters):
        for c in range(C):
            plans[c].matvec(P[c], rho, lams[c], out=Q[c])
            vec(c)


M, V = torch.cuda.Stream(), torch.cuda.Stream()


def pipelined():
    cur = torch.cuda.current_stream()
    M.wait_stream(cur), V.wait_stream(cur)
    evM = [None] * C
    evV = [None] * C
    for k in range(iters):
        for c in range(C):
            with torch.cuda.stream(M):
                if evV[c] is not None:
                    M.wait_event(evV[c])
                plans[c].matvec(P[c], rho, lams[c], out=Q[c])
                evM[c] = torch.cuda.Event()
                evM[c].record(M)
            with torch.cuda.stream(V):
                V.wait_event(evM[c])
                vec(c)
                evV[c] = torch.cuda.Event()
                evV[c].record(V)
    cur.wait_stream(M), cur.wait_stream(V)


def timed(f, rep=3):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rep):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6 / (iters * C)


out = {'workload': name, 'channels': C, 'iters': iters, 'env': {k: v for k, v in os.environ.items() if k.startswith('UNIRES_')}}
out['serial_us_per_channel_iteration'] = timed(serial)
out['pipelined_us_per_channel_iteration'] = timed(pipelined)
print(json.dumps(out), flush=True)
# graph replay of both (no host launch gaps) - GRAPH=1
for tag, f in ((('serial', serial), ('pipelined', pipelined)) if os.environ.get('GRAPH') else ()):
    try:
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            f()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=s):
                f()
        torch.cuda.synchronize()
        out[tag + '_graph_us_per_channel_iteration'] = timed(gr.replay)
    except Exception as e:  # noqa
        out[tag + '_graph_error'] = str(e)[:200]
print(json.dumps(out))
