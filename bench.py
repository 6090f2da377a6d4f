#!/usr/bin/env python
"""bench.py - the UniRes y-update hot path on N MI355X GPUs (one process per GPU).

A "step" is one full y-update of one synthetic subject (BASELINE.json configs[2]:
256^3, 3 channels, 6 mm thick slices along z, small random rigid per channel):
for every channel assemble b = tau At x - lam Dt(w - rho z) and run 20 CG
iterations of (tau AtA + rho lam^2 DtD) y = b in fixed-iteration mode
(tolerance 0 - no early stop, so the work per step is constant).

metric  = per-channel CG iterations per second, whole job (all ranks).
roofline = the CG matvec (AtA + rho lam^2 DtD) for one channel, algorithmic
           bytes B_mv = 4*(2 N_y + 2 N_x) / measured duration.
cpu_baseline = the CPU oracle (torch-CPU restatement of the reference path)
           timed on a bounded sample of the same workload on this box's cores.

Multi-GPU: subjects are independent: rank g reconstructs its own
subject, no data-path collective; RCCL is used for the start/end barrier and
the max-over-ranks of the elapsed time only.  scaling = "weak".
"""
import argparse
import json
import math
import os
import sys
import time

import torch

import unires_amd  # noqa: F401  (first: its import sets ROC_SIGNAL_POOL_SIZE before anything initialises the HIP runtime)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy

from workloads import WORKLOADS, build_subject, orient_axes, phantom, rigid_matrix  # noqa: E402,F401


def alg_bytes_matvec(x_c, dim_y, do_proj=True):
    """B_mv of SURVEY 8(d): 4 (2 N_y + 2 sum N_x) for R1/R2, 4 * 2 N_y for R0 (A = I)."""
    n_y = dim_y[0] * dim_y[1] * dim_y[2]
    if not do_proj:
        return 4 * 2 * n_y
    n_x = sum(xn.po.dim_x[0] * xn.po.dim_x[1] * xn.po.dim_x[2] for xn in x_c)
    return 4 * (2 * n_y + 2 * n_x)


def pmc_traffic(workload):
    """HBM bytes per matvec launch from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs; gfx950 FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes).  The figure is STATIC - read from a committed profile, not
    measured in this run (PMC passes need rocprofv3 around the process) - and is tagged with the
    profile file and its git blob hash so that it can be checked against the tree.
    None if no profile of this workload is committed."""
    import hashlib
    for name in ('r06_traffic.json', 'r05_traffic.json', 'r04_traffic.json', 'r03_traffic.json', 'r02_traffic.json', 'r01_traffic_aligned.json'):
        path = os.path.join(ROOT, 'profiles', name)
        try:
            raw = open(path, 'rb').read()
            rec = json.loads(raw)
            if rec.get('workload') == workload:
                blob = hashlib.sha1(b'blob %d\0' % len(raw) + raw).hexdigest()
                return {'bytes_per_launch': rec['bytes_per_launch'], 'static': True,
                        'source': 'profiles/' + name, 'git_blob': blob,
                        'note': 'PMC passes of the kernels as committed with that profile; re-run '
                                'tools/refresh_profiles_r06.sh to refresh'}
        except (OSError, ValueError, KeyError):
            pass
    return None


def time_matvec(x, y, rho, sett, reps=16, ring=2, graph=True, channels=None):
    """Average duration of one CG matvec (mean over the subject's channels, whose rigid
    transforms - and therefore kernel costs - differ), HIP events on the launch stream.
    The launches cycle through channels and ``ring`` distinct (p, q) pairs per channel
    (12 x 67 MB at 256^3 x 3, more than the 256 MB Infinity Cache), as inside CG where p was
    just rewritten and other vectors were streamed in between: timing one p/q pair back to
    back reads p from the cache and comes out faster than the same kernel does in the solver
    (aligned kernel: 35.7 us hot, 42 us cold)."""
    from unires_amd._project import _channel_plan
    dev = y[0].dat.device
    chans = list(range(len(x))) if channels is None else list(channels)
    C = len(chans)
    plans = [_channel_plan(x[c], y[c], sett.method, sett.do_proj) for c in chans]
    ps = [[torch.rand(y[c].dim, device=dev) for _ in range(ring)] for c in chans]
    qs = [[torch.empty_like(ps[k][0]) for _ in range(ring)] for k in range(C)]

    def sweep(n):
        for i in range(n):
            for k, c in enumerate(chans):
                plans[k].matvec(ps[k][i % ring], rho, y[c].lam, out=qs[k][i % ring])

    sweep(ring + 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph:
        # the launches replayed as one hipGraph, as unires_cg_solve runs them: without it every
        # kernel of the pair waits ~4 us for its launch and the figure is 5 % above the kernels'
        # own durations (rocprofv3)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sweep(reps)
            g.replay()
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / (reps * C)
        except Exception as exc:  # capture refused: time the plain launches
            sys.stderr.write('matvec timing: graph capture failed (%s), timing eager launches\n' % exc)
            torch.cuda.synchronize()
    e0.record()
    sweep(reps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * C)


def time_matvec_in_solve(x, y, z, w, rho, tmp, sett, serial=True):
    """Average duration of one operator application A(p) INSIDE the CG solves of one y-update: the
    library brackets each of them with HIP events on the stream it launches on
    (unires_plan_time_matvecs; the solves then run as plain launches - a dependent kernel boundary costs
    the same in a hipGraph).  This is the kernel pair as the timed region runs it - p freshly rewritten
    by the preceding update, the other vectors streamed in between - and the figure the rocprofv3 kernel
    durations of this command, dominated by the solves' own launches, agree with."""
    import unires_amd as U
    from unires_amd._project import _channel_plan
    plans = [_channel_plan(x[c], y[c], sett.method, sett.do_proj) for c in range(len(x))]
    # serial: one channel after the other, whatever the setting - with the channels on separate streams several
    # kernels share the chip and a launch's wall time is not the kernel's own (that figure, `serial=False`, is
    # reported beside it: a launch takes LONGER under sharing while the job as a whole gets faster)
    keep = getattr(sett, 'channel_streams', 'auto')
    if serial:
        sett.channel_streams = False
    for yc in y:
        yc.dat.zero_()
    U._update_y(x, y, z, w, rho, tmp, sett)  # (warm: plans, caches, allocator)
    torch.cuda.synchronize()
    for pl in plans:
        pl.time_matvecs(True)
    try:
        for yc in y:
            yc.dat.zero_()
        U._update_y(x, y, z, w, rho, tmp, sett)
        torch.cuda.synchronize()
        n, us = 0, 0.0
        for pl in plans:
            k, t = pl.matvec_time()
            n, us = n + k, us + t
    finally:
        for pl in plans:
            pl.time_matvecs(False)
        sett.channel_streams = keep
    return us * 1e-6 / max(n, 1), n


def time_matvec_in_graph(x, y, z, w, rho, tmp, sett, reps=10):
    """What one operator application costs inside the solve AS PRODUCTION RUNS IT: a fixed-iteration solve is one
    hipGraph, replayed (api.hip) - no host launch latency between its kernels, which the event-bracketed plain
    launches of `time_matvec_in_solve` pay (17 us for a 13 us kernel at 181 x 217 x 181).  Event records do not
    survive stream capture on this runtime (tools/mb_graph_event.hip: the nodes are dropped), so the figure is a
    difference: y-updates whose solves enqueue every A(p) twice (unires_plan_time_matvecs(plan, 2): idempotent, same
    result, still one graph) against plain ones, one channel after the other, median of `reps` each, divided by the
    number of operator applications added.  Returns (seconds per application, y-update plain, y-update doubled)."""
    import statistics
    import unires_amd as U
    from unires_amd._project import _channel_plan
    plans = [_channel_plan(x[c], y[c], sett.method, sett.do_proj) for c in range(len(x))]
    keep = getattr(sett, 'channel_streams', 'auto')
    sett.channel_streams = False

    def run(mode):
        for pl in plans:
            pl.time_matvecs(mode)
        ts = []
        for rep in range(reps + 2):
            for yc in y:
                yc.dat.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            U._update_y(x, y, z, w, rho, tmp, sett)
            e1.record()
            torch.cuda.synchronize()
            if rep >= 2:  # (the first replays capture / warm)
                ts.append(e0.elapsed_time(e1) * 1e-3)
        return statistics.median(ts)
    try:
        t1 = run(False)
        t2 = run(2)
    finally:
        for pl in plans:
            pl.time_matvecs(False)
        sett.channel_streams = keep
    return (t2 - t1) / (len(x) * sett.cgs_max_iter), t1, t2


def host_cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(wl, seconds_budget=25.0, device=None):
    """CPU oracle ("port" of the reference composition) on one channel of the same
    workload: fixed-iteration CG, as many iterations as fit the budget (>= 1).

    The oracle does not scale with threads (its trilinear push is eight ``index_add_`` passes over the
    whole grid: 128 threads are slower than one), so the thread count is CHOSEN first: one matvec of
    the same operator at half the linear size is timed at 1, 8, 16, 32, 64 and all host threads, and the
    full-size sample then runs at the fastest of them - ``cores`` is that count, ``thread_scan`` holds
    all of them.

    The first full-size oracle matvec is also the full-size parity check of SURVEY 8(d): the HIP matvec
    runs on the SAME operator and the SAME input and the float32 relative / max-abs error is
    reported next to the timing (gate 1e-4)."""
    from oracle import nitorch_restated as N
    from tests.helpers import matvec_parity, oracle_channel, oracle_lhs  # (test infrastructure, like oracle/)
    dim_y = wl['dim_y']
    keep_threads = torch.get_num_threads()  # (batch.init_from_env caps the rank's threads: the baseline gets the host)
    all_threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    # ---- thread scan on the half-size problem (a fraction of a second per point) ----
    small = tuple(max(16, d // 2) for d in dim_y)
    Ps = oracle_channel(wl, small, seed=1)
    lhs_s = oracle_lhs(wl, Ps)
    nvox_s = small[0] * small[1] * small[2]
    scan = {}
    try:
        # (beyond 64 threads the oracle's index_add_ passes collapse - 42 s per half-size matvec at 256 - and the
        # scan would take longer than the sample it sizes: the host's full count is scanned only up to 64)
        for nt in sorted(set(t for t in (1, 8, 16, 32, 64, min(all_threads, 64)) if t <= all_threads)):
            torch.set_num_threads(nt)
            lhs_s(Ps['b'])  # warm
            t0 = time.perf_counter()
            lhs_s(Ps['b'])
            scan[nt] = time.perf_counter() - t0
    finally:
        torch.set_num_threads(keep_threads)
    best = min(scan, key=scan.get)
    # ---- the bounded full-size sample at the fastest thread count ----
    P = oracle_channel(wl, dim_y)
    lhs = oracle_lhs(wl, P)
    torch.set_num_threads(best)
    try:
        t0 = time.perf_counter()
        q_cpu = lhs(P['b'])  # one matvec: sizes the sample AND is the parity reference
        t_mv = time.perf_counter() - t0
        parity = matvec_parity(wl, P, q_cpu, device) if device is not None else None
        n_it = max(1, min(20, int(seconds_budget / max(t_mv, 1e-3)) - 1))
        t0 = time.perf_counter()
        N.cg(lhs, P['b'], P['yc'].dat, max_iter=n_it, tolerance=0, stop='max_gain')
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(keep_threads)
    nvox = dim_y[0] * dim_y[1] * dim_y[2]
    # cg(tolerance=0) does n_it + 1 matvecs for n_it iterations; report iterations/s
    out = dict(value=n_it / dt, unit='cg_iters/s', cores=best, kind='port', cpu_model=host_cpu_model(),
               host_threads=all_threads,
               sample='%d CG iterations (tol=0) of one %dx%dx%d channel, oracle/unires_restated '
                      '(torch-CPU, unfused as the reference composes it) on %d threads - the fastest of '
                      'the scanned counts; matvec %.2f s' % (n_it, dim_y[0], dim_y[1], dim_y[2], best, t_mv),
               matvec_s=t_mv, matvec_Mvox_per_s=nvox / t_mv / 1e6,
               thread_scan={'sample': 'one matvec of a %dx%dx%d channel (same operator shape)' % small,
                            'matvec_s': {str(k): v for k, v in scan.items()},
                            'matvec_Mvox_per_s': {str(k): nvox_s / v / 1e6 for k, v in scan.items()}},
               single_thread={'cores': 1, 'matvec_s': scan.get(1), 'matvec_Mvox_per_s': nvox_s / scan[1] / 1e6,
                              'sample': 'one matvec of a %dx%dx%d channel (same operator shape)' % small})
    if parity is not None:
        out['parity'] = parity
    return out


def time_steps(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def variants(name, x, y, z, w, rho, tmp, sett, device):
    """Secondary records of the same y-update (rank 0, N = 1; never the headline value):
    (1) reference-faithful CG: tolerance 1e-3, stop 'max_gain' (objective every iteration,
        early stop on device) with the realised iteration counts;
    (2) the same subject with rigid = identity (SURVEY 8(d) allows "small random rigid or
        identity"): observations on the reconstruction grid take the one-kernel matvec."""
    import unires_amd as U
    out = {}

    def run(xs, ys, zs, ws, r, t, st):
        def step():
            for yc in ys:
                yc.dat.zero_()
            U._update_y(xs, ys, zs, ws, r, t, st)
        return step

    tol0 = sett.cgs_tol
    sett.cgs_tol = 1e-3
    for yc in y:
        yc.dat.zero_()
    info = []
    U._update_y(x, y, z, w, rho, tmp, sett, info=info)
    iters = [int(r[0]) for r in info]
    # every step timed on its own (device idle before and after), 3 warm-ups, 12 steps: median + spread
    step = run(x, y, z, w, rho, tmp, sett)
    ts = []
    for i in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    t = ts[len(ts) // 2]
    out['cg_tol1e-3_max_gain'] = {'cg_iters_realised': iters, 'cg_iters_per_sec': sum(iters) / t,
                                  'ms_per_step': t * 1e3, 'ms_per_step_min': ts[0] * 1e3,
                                  'ms_per_step_max': ts[-1] * 1e3, 'steps_timed': len(ts),
                                  'ms_per_realised_iteration': t * 1e3 / max(1, sum(iters)),
                                  'timing': 'median of 12 y-updates, each between device synchronisations, after 3 '
                                            'warm-ups; y-update = C x (RHS + CG to the reference-default stopping rule)'}
    sett.cgs_tol = tol0
    alt = name + '_aligned'
    if alt in WORKLOADS:
        del info
        xa, ya, za, wa, rhoa, setta = build_subject(WORKLOADS[alt], device, seed=1234)
        ta = torch.zeros_like(ya[0].dat)
        t = time_steps(run(xa, ya, za, wa, rhoa, ta, setta), 3, 1)
        t_mv = time_matvec(xa, ya, rhoa, setta, graph=False)
        b_mv = alg_bytes_matvec(xa[0], WORKLOADS[alt]['dim_y'])
        out[alt] = {'cg_iters_per_sec': len(xa) * setta.cgs_max_iter / t, 'ms_per_step': t * 1e3,
                    'matvec_us': t_mv * 1e6, 'matvec_GBps': b_mv / t_mv / 1e9,
                    'matvec_frac_of_peak': b_mv / t_mv / 1e9 / HBM_PEAK_GBS}
    return out


def spawn_ranks(n, argv, dry_run=False):
    """`python bench.py --gpus N` without a launcher: run N ranks (one per GPU, RCCL) through
    torch.distributed.run on 127.0.0.1, exactly as the driver's command line does."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    if dry_run:
        return cmd
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='cfg3_256c3_thick6z', choices=list(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variants', action='store_true')
    ap.add_argument('--serial-channels', action='store_true',
                    help='run the channels of the y-update one after the other on one stream '
                         '(profiling aid: per-kernel durations then carry no overlap)')
    ap.add_argument('--channel-streams', action='store_true',
                    help='the channels of the y-update on separate HIP streams, whatever the volume size '
                         "(settings.channel_streams = 'auto' picks streams below 10 M voxels)")
    ap.add_argument('--no-tol-leg', action='store_true',
                    help='skip the subject leg under the reference-default stopping rule (profiling aid: its solves '
                         'enqueue launches that return at entry once a solve has stopped, which deflate the per-kernel '
                         'averages of a rocprofv3 --stats summary)')
    ap.add_argument('--cpu-seconds', type=float, default=25.0)
    ap.add_argument('--admm-iters', type=int, default=50,
                    help='ADMM iterations of the subjects/sec leg (one subject = this many iterations)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started as plain `python bench.py --gpus N`: re-launch as N ranks, one per GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    from unires_amd import batch
    import torch.distributed as td
    rank, world, local_rank = batch.init_from_env(backend='nccl')
    dist = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if dist:
        td.barrier()
    import unires_amd as U

    wl = WORKLOADS[args.workload]
    # every rank reconstructs a subject of identical cost (same geometry draw): the kernels'
    # run time depends on the rigid transforms, and weak scaling should not measure that spread
    x, y, z, w, rho, sett = build_subject(wl, device, seed=1234)
    if args.serial_channels:
        sett.channel_streams = False
    if args.channel_streams:
        sett.channel_streams = True
    tmp = torch.zeros_like(y[0].dat)

    def step():
        for yc in y:  # same start every step -> identical work
            yc.dat.zero_()
        U._update_y(x, y, z, w, rho, tmp, sett)

    # (plans, schedules and the solves' hipGraphs are built by the first steps, and the second replay of a fresh graph
    # still costs ~30 ms on this runtime: never fewer than three untimed steps, whatever --warmup says)
    for _ in range(max(args.warmup, 3)):
        step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    C = wl['C']
    iters_per_step = C * sett.cgs_max_iter
    total_iters = world * args.steps * iters_per_step
    # subjects/sec: EVERY rank runs n_admm full ADMM iterations (y-update C x 20 CG,
    # objective, z- and w-update) of its own subject between barriers; time = max over ranks
    n_admm = max(1, args.admm_iters)
    sett.tolerance = 1e-4

    def subject_leg(cgs_tol):
        """One subject from a zero start (y, z, w): warm-up iteration, then n_admm timed ones.  Returns
        (max over ranks, this rank's own time, every rank's time)."""
        keep = sett.cgs_tol
        sett.cgs_tol = cgs_tol
        try:
            for yc in y:
                yc.dat.zero_()
            z.zero_(), w.zero_()
            obj = torch.zeros((n_admm + 1, 3), dtype=torch.float64, device=device)
            U._update_admm(x, y, z, w, rho, tmp, obj, 0, sett)  # warm-up (plans, graphs)
            torch.cuda.synchronize()
            if dist:
                td.barrier()
                torch.cuda.synchronize()
            ta = time.perf_counter()
            for it in range(1, n_admm + 1):
                U._update_admm(x, y, z, w, rho, tmp, obj, it, sett)
            torch.cuda.synchronize()
            own = time.perf_counter() - ta  # this rank's own subject, before it waits for the others
            if dist:
                td.barrier()
                torch.cuda.synchronize()
            t_all = time.perf_counter() - ta
            per_rank = [own]
            if dist:
                t = torch.tensor([t_all], dtype=torch.float64, device=device)
                td.all_reduce(t, op=td.ReduceOp.MAX)
                t_all = float(t.item())
                g = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
                td.all_gather(g, torch.tensor([own], dtype=torch.float64, device=device))
                per_rank = [float(v.item()) for v in g]
            return t_all, own, per_rank
        finally:
            sett.cgs_tol = keep

    t_subject, _, t_subject_ranks = subject_leg(0.0)
    # the same subject with the reference's DEFAULT solver settings (struct.py:65-67: cgs_tol = 1e-3,
    # 'max_gain'): the solves stop where the reference's would
    t_subject_tol, _, t_subject_tol_ranks = subject_leg(1e-3) if not args.no_tol_leg else (None, None, [])
    out = None
    if rank == 0:
        # the headline figure is ONE fixed method (r1's): plain launches on the stream, HIP events around
        # them - the figure that must agree with the rocprofv3 kernel durations committed under
        # profiles/ (kernel sum 123.3 us vs 123 - 125 us by events; a hipGraph replay of the same launches
        # carries ~2.5 us of barrier packet per kernel and is reported next to it, never mixed in)
        t_mv_graph = time_matvec(x, y, rho, sett)
        t_mv_eager = time_matvec(x, y, rho, sett, graph=False)
        # (r3) the headline figure is the operator application as the timed region runs it: every A(p) of
        # one y-update's CG solves between HIP events recorded by the library on its launch stream.  Since
        # the iterate x is streamed past the caches the solver's matvecs find p warm, and the cold-operand
        # figure (kept as us_per_launch_cold) came out ~5 % above the kernel durations rocprofv3 reports
        # for this command.
        t_mv, n_mv = time_matvec_in_solve(x, y, z, w, rho, tmp, sett)
        t_mv_ingraph, t_yu1, t_yu2 = time_matvec_in_graph(x, y, z, w, rho, tmp, sett)
        streams_on = bool(U._update.channel_streams_on(sett, y[0].dat)) and C > 1
        t_mv_shared = time_matvec_in_solve(x, y, z, w, rho, tmp, sett, serial=False)[0] if streams_on else None
        # ... and the job the other way round: the same steps with one channel after the other
        value_serial = None
        if streams_on and world == 1:
            keep_cs = sett.channel_streams
            sett.channel_streams = False
            try:
                # (three warm-up steps: the plans leave the shared-chip mode, their solves are captured anew, and the
                # SECOND replay of a fresh graph still costs ~30 ms on this runtime - one warm-up step read 3 100 it/s)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                ts = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                value_serial = args.steps * iters_per_step / (time.perf_counter() - ts)
            finally:
                sett.channel_streams = keep_cs
        per_channel = [time_matvec(x, y, rho, sett, ring=4, channels=[c], graph=False) * 1e6 for c in range(len(x))]
        b_mv = alg_bytes_matvec(x[0], wl['dim_y'], sett.do_proj)
        achieved = b_mv / t_mv / 1e9
        traffic = pmc_traffic(args.workload)
        out = {
            'metric': 'cg_iters_per_sec', 'value': total_iters / elapsed, 'unit': 'cg_iters/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': args.workload, 'dim_y': list(wl['dim_y']), 'channels': C,
                       'thick_ratio': wl['thick'], 'cg_iters_per_channel': sett.cgs_max_iter,
                       'cg_mode': 'fixed-iteration (tol=0), identity preconditioner',
                       'step': 'one y-update of one subject: C x (RHS + 20 CG iterations)',
                       'channel_streams': bool(U._update.channel_streams_on(sett, y[0].dat)),
                       'channel_streams_setting': getattr(sett, 'channel_streams', 'auto'),
                       'parallelism': 'one subject per GPU, no data-path collective'},
            # `value` with one channel after the other (settings.channel_streams = False), same steps
            'value_channels_serial': value_serial,
            'subjects_per_sec': world / t_subject,
            'subjects_per_sec_note': 'subject = %d full ADMM iterations (y-update C x 20 CG, objective, z- and '
                                     'w-update) run on every rank between barriers, max over ranks: %.3f s '
                                     '(%.2f ms per ADMM iteration)' % (n_admm, t_subject, t_subject / n_admm * 1e3),
            # the reference's default solver settings (cgs_tol = 1e-3, stop 'max_gain', struct.py:65-67)
            'subjects_per_sec_tol1e-3': world / t_subject_tol if t_subject_tol else None,
            'subjects_per_sec_tol1e-3_note': 'the same %d ADMM iterations with every CG solve stopping where the '
                                             "reference's does (tolerance 1e-3 on the gain of the objective): %.3f s "
                                             '(%.2f ms per ADMM iteration)' % (n_admm, t_subject_tol or 0.0,
                                                                             (t_subject_tol or 0.0) / n_admm * 1e3),
            # every rank's own time for its subject, before the closing barrier (s): a scaling run explains itself
            't_subject_per_rank': {'min': min(t_subject_ranks), 'max': max(t_subject_ranks), 'all': t_subject_ranks},
            't_subject_tol1e-3_per_rank': ({'min': min(t_subject_tol_ranks), 'max': max(t_subject_tol_ranks),
                                            'all': t_subject_tol_ranks} if t_subject_tol_ranks else None),
            'roofline': {'bound': 'hbm', 'kernel': 'ata_matvec (per launch inside the CG solves of one y-update, mean over channels)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         # HBM bytes per launch from the PMC counters (null if no profile of this workload is
                         # committed); STATIC: read from the committed profile named in traffic_source
                         'traffic': (traffic or {}).get('bytes_per_launch'), 'traffic_source': traffic,
                         'alg_bytes_per_launch': b_mv, 'us_per_launch': t_mv * 1e6,
                         'launches_timed': n_mv,
                         # the same launches as the TIMED REGION runs them when the channels share the chip
                         # (streams of their own, persistent grids sized for it): longer launches, a faster job
                         'us_per_launch_channels_overlapped': None if t_mv_shared is None else t_mv_shared * 1e6,
                         # ... and inside the replayed hipGraph that a fixed-iteration solve is in production (no host
                         # launch latency): (y-update with every A(p) enqueued twice - plain y-update) / applications
                         'us_per_launch_in_graph': t_mv_ingraph * 1e6,
                         'frac_in_graph': b_mv / t_mv_ingraph / 1e9 / HBM_PEAK_GBS,
                         'in_graph_y_update_ms': [t_yu1 * 1e3, t_yu2 * 1e3],
                         'us_per_launch_cold': t_mv_eager * 1e6, 'frac_cold': b_mv / t_mv_eager / 1e9 / HBM_PEAK_GBS,
                         'us_per_launch_graph': t_mv_graph * 1e6, 'us_per_launch_eager': t_mv_eager * 1e6,
                         'us_per_launch_by_channel': per_channel,
                         'timing': 'us_per_launch: HIP events recorded by the library around every A(p) of one '
                                   "y-update's CG solves, on the stream they are launched on (plain launches), one "
                                   'channel after the other - the kernels alone on the chip; '
                                   '_in_graph: marginal cost of an A(p) inside the hipGraph a tol = 0 solve is '
                                   'replayed as (doubled-matvec solves minus plain ones, medians of 10); '
                                   '_channels_overlapped: the same with the channels on streams of their own, as '
                                   "`value`'s timed region runs them; "
                                   '_cold = _eager: stand-alone launches cycling through more p / q buffers than the '
                                   'Infinity Cache holds (the round-2 method), _graph: those replayed as one hipGraph, '
                                   '_by_channel: cold, one channel at a time'},
        }
        if world == 1 and not args.no_variants:
            out['variants'] = variants(args.workload, x, y, z, w, rho, tmp, sett, device)
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(wl, args.cpu_seconds, device=device)
        print(json.dumps(out))
        sys.stdout.flush()
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == '__main__':
    main()
