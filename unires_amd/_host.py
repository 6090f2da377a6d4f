"""Host-side housekeeping of the one-process-per-GPU batch mode: how many CPU threads a
rank may use, which cores it sits on, and how it waits for its GPU.

Eight ranks share one host.  Left alone, every rank (i) lets torch / BLAS fan tiny host-side algebra (4 x 4
solves, 6 x 6 Gauss-Newton systems) out over all cores, (ii) runs ahead of its GPU until the hardware queue is full and then SPINS in
every launch call, and (iii) spins in every device synchronisation - `host_cpu_ms_per_iteration` 14.2 for a
13.7 ms ADMM iteration (profiles/r04_host_time.jsonl): a core per rank burnt on waiting.  Nothing of this
is the reference's (single process, `unires/run.py`); it is what the batch mode adds around it."""
import os
import time

import torch


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-')
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def _gpu_local_cpus(local_rank):
    """CPUs of the NUMA node the rank's GPU hangs off (sysfs ``local_cpulist`` of its PCI function), or None."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        dom, bus, devn = getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id
        path = '/sys/bus/pci/devices/%04x:%02x:%02x.0/local_cpulist' % (dom, bus, devn)
        with open(path) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except Exception:
        return None


def core_share(cpus, world, local_rank):
    """The slice of ``cpus`` rank ``local_rank`` of ``world`` gets: contiguous, equal, at least one core."""
    cpus = sorted(cpus)
    n = max(1, len(cpus) // max(1, world))
    lo = (local_rank % max(1, len(cpus) // n)) * n
    return cpus[lo:lo + n] or cpus[:1]


def configure_host(local_rank=0, world=1, gpu_numa=True):
    """Cap this rank's torch intra-op threads at its share of the cores (never more than 8: the package's host
    work is tiny matrices) and pin the process to a core set of its own - the GPU's NUMA node first where sysfs
    exposes it, then split among the ranks - so that eight ranks neither fight over the same cores nor
    migrate.  Environment: UNIRES_HOST_THREADS=<n> overrides the thread cap, UNIRES_CPU_AFFINITY=0 leaves the
    affinity alone.  Returns dict(threads, cpus) of what was applied."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:  # not Linux
        avail = list(range(os.cpu_count() or 1))
    share = max(1, len(avail) // max(1, world))
    threads = min(share, 8)
    if os.environ.get('UNIRES_HOST_THREADS'):
        threads = max(1, int(os.environ['UNIRES_HOST_THREADS']))
    torch.set_num_threads(threads)
    cpus = None
    if world > 1 and os.environ.get('UNIRES_CPU_AFFINITY', '1') != '0' and hasattr(os, 'sched_setaffinity'):
        pool = avail
        if gpu_numa and torch.cuda.is_available():
            local = _gpu_local_cpus(local_rank)
            if local:
                local = [c for c in local if c in set(avail)]
                # the ranks whose GPUs share this node split it; without knowing them, split by world
                if len(local) >= world:
                    pool = local
        cpus = core_share(pool, world, local_rank)
        try:
            os.sched_setaffinity(0, set(cpus))
        except OSError:
            cpus = None
    return dict(threads=threads, cpus=cpus)


_capped = False


def cap_threads(n=8):
    """Cap torch's intra-op threads at ``n`` ONCE per process (UNIRES_HOST_THREADS overrides; a process that
    wants more sets it after).  The package's host work is tiny matrices (4 x 4 solves, matrix exponentials,
    6 x 6 Gauss-Newton systems): nothing to gain from a 256-thread pool, and eight ranks' pools would fight.
    Not a context manager on purpose: switching the pool's size around every call is work of its own.  (The
    "32 ms of CPU for a 5 ms rigid step" of profiles/r04_fit.jsonl turned out to be wall + three scheduler
    ticks charged to runtime helper threads, tools/rigid_threads.py - not this.)"""
    global _capped
    if _capped:
        return
    _capped = True
    if os.environ.get('UNIRES_HOST_THREADS'):
        n = max(1, int(os.environ['UNIRES_HOST_THREADS']))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)


_blas_ctl = None


def _blas_one_thread():
    """Context: numpy / scipy BLAS (OpenBLAS / MKL pools, which spin after every call) on ONE thread - the 4 x 4
    `expm` / `logm` / 6 x 6 solves of the Gauss-Newton steps would wake a pool of as many threads as the host has
    cores.  threadpoolctl where it is installed
    (the controller is made once: introspecting the loaded libraries costs milliseconds), else nothing."""
    global _blas_ctl
    if _blas_ctl is None:
        try:
            from threadpoolctl import ThreadpoolController
            _blas_ctl = ThreadpoolController()
        except Exception:
            _blas_ctl = False
    if _blas_ctl:
        return _blas_ctl.limit(limits=1, user_api='blas')
    import contextlib
    return contextlib.nullcontext()


def light_host(fn):
    """Decorator: ``fn`` is a host section made of tiny matrices - see `cap_threads`, `_blas_one_thread`."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        cap_threads()
        with _blas_one_thread():
            return fn(*a, **k)
    return wrapped


def _sleep_until(ev, dt=1e-4):
    """Wait for a recorded event by polling it between SLEEPS.  Measured on the MI355X box (tools/wait_probe.py
    -> profiles/r05_wait_probe.txt): `torch.cuda.synchronize()` and even `Event(blocking=True).synchronize()`
    burn a core for as long as they wait unless the process called hipSetDeviceFlags(hipDeviceScheduleBlockingSync)
    before its context existed (torch has none of that); query + sleep costs 1 % of a core whatever the flags
    and ends within ``dt`` of the event."""
    while not ev.query():
        time.sleep(dt)


class Pacer:
    """Keeps the host at most ``depth`` steps ahead of the device, SLEEPING while it waits.

    Without it a loop of enqueue-only steps (the ADMM iterations: no read-back in between) fills the hardware
    queue and every further launch call spins inside the runtime until there is room."""

    def __init__(self, depth=2):
        self.depth = max(1, int(depth))
        self._events = []

    def step(self, stream=None):
        """Call once per enqueued step, after its last launch."""
        if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return
        if len(self._events) >= self.depth:
            _sleep_until(self._events.pop(0), 5e-4)  # (an event `depth` steps old: no hurry)
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        self._events.append(ev)

    def drain(self):
        while self._events:
            _sleep_until(self._events.pop(0))


def wait_blocking(device=None):
    """Wait for everything enqueued on the current stream WITHOUT spinning: what `tensor.cpu()` /
    `torch.cuda.synchronize()` do by polling flat out."""
    if not torch.cuda.is_available():
        return
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _sleep_until(ev, 5e-5)
