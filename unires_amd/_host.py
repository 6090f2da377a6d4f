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


def tune_runtime():
    """ROC_SIGNAL_POOL_SIZE=4096 in the environment unless the user set it (or UNIRES_NO_RUNTIME_TUNING=1) - to
    take effect it must be there before the process's first HIP call, hence at import.

    Why: the HIP runtime hands every packet that carries a completion signal (markers, cross-stream waits,
    graph launches, read-backs) a signal from a ring of ROC_SIGNAL_POOL_SIZE (default 64) per hardware queue.
    An ADMM iteration enqueues ~10^3 commands ahead of the device; with the ring that short the runtime's
    signal thread spends its time - system time, almost no voluntary context switches - waiting on recycled
    signals: 7.4 ms of CPU per 19.2 ms iteration with the reference's stopping rule at 256^3, 6.0 of 6.9 ms
    with the three channels of a 181^3 subject on streams of their own (tools/host_profile.py,
    profiles/r05_host_profile.txt).  With 4096 signals that thread is idle (0.1 - 0.3 ms), wall times and
    results unchanged.  Nothing here is the reference's; it is what eight ranks sharing one host need."""
    if os.environ.get('UNIRES_NO_RUNTIME_TUNING') or 'ROC_SIGNAL_POOL_SIZE' in os.environ:
        return
    os.environ['ROC_SIGNAL_POOL_SIZE'] = '4096'
    if torch.cuda.is_initialized():
        # (an embedding application that touched the GPU first: the variable is read once, at the runtime's start)
        import warnings
        warnings.warn(
            'unires_amd was imported after the HIP runtime was initialised: ROC_SIGNAL_POOL_SIZE=4096 cannot take '
            'effect in this process.  Results are unaffected; the runtime\'s signal thread then costs ~6-7 ms of host '
            'CPU per ADMM iteration and channel streams lose ~7 % at 256^3 (profiles/r05_host_profile.txt, '
            'r05_overlap_scan.txt).  Export ROC_SIGNAL_POOL_SIZE=4096 before the process starts, or import unires_amd '
            'before the first CUDA/HIP call (INTEGRATION.md 2); UNIRES_NO_RUNTIME_TUNING=1 silences this.',
            RuntimeWarning, stacklevel=3)


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
    global batch_mode
    share = max(1, len(avail) // max(1, world))
    threads = min(share, 8) if world > 1 else torch.get_num_threads()  # (a lone process keeps its pool)
    if os.environ.get('UNIRES_HOST_THREADS'):
        threads = max(1, int(os.environ['UNIRES_HOST_THREADS']))
    if threads != torch.get_num_threads():
        torch.set_num_threads(threads)
    batch_mode = batch_mode or world > 1
    cpus = None
    if world > 1 and os.environ.get('UNIRES_CPU_AFFINITY', '1') != '0' and hasattr(os, 'sched_setaffinity'):
        pool = avail
        if gpu_numa and torch.cuda.is_available():
            # (the rank's device: LOCAL_RANK, or device 0 when HIP_VISIBLE_DEVICES gives every rank one GPU)
            local = _gpu_local_cpus(local_rank if local_rank < torch.cuda.device_count() else 0)
            if local:
                local = [c for c in local if c in set(avail)]
                # the ranks whose GPUs share this node split it; without knowing them, split by world
                if len(local) >= world:
                    pool = local
        cpus = core_share(pool, world, local_rank)
        try:
            _pin_all_threads(set(cpus))
        except OSError:
            cpus = None
    return dict(threads=threads, cpus=cpus)


def _pin_all_threads(cpus):
    """`sched_setaffinity` on EVERY thread of the process: reading the GPU's PCI address above started the HIP
    runtime, whose helper threads (the ones the host profiles blame for CPU burn) exist by now and would keep the
    old mask - `sched_setaffinity(0, ...)` pins the calling thread and those created after it only."""
    os.sched_setaffinity(0, cpus)
    try:
        tids = [int(t) for t in os.listdir('/proc/self/task')]
    except OSError:
        return
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
        except OSError:  # (a thread that ended meanwhile)
            pass


# True once this process is one rank of several on the host (configure_host with world > 1) or when
# UNIRES_LIGHT_HOST=1: only then do the host sections below shrink the process-wide thread pools for good
batch_mode = os.environ.get('UNIRES_LIGHT_HOST', '0') == '1'


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


_blas_limit = None


def cap_blas(n=1):
    """numpy / scipy BLAS (OpenBLAS pthread pools) on ``n`` threads, ONCE per process and for good
    (UNIRES_BLAS_THREADS overrides, 0 = leave the pools alone).  The Gauss-Newton steps run 4 x 4 `expm` / `logm`
    and 6 x 6 solves through numpy: on a 256-core host OpenBLAS keeps a pool of 64 workers per loaded copy, a few of
    which it wakes even for these sizes - and a woken worker spins for ~100 ms before it sleeps again, i.e. for ever
    when a step comes every 5 ms.  Measured on the MI355X box (tools/rigid_profile.py, profiles/r05_rigid_profile.txt):
    three workers at 100 % next to the main thread, 20 ms of CPU for a 5 ms rigid step; 5.2 ms with the pools on one
    thread.  Sticky on purpose: limiting around each call (threadpoolctl as a context manager, round 5's first
    attempt) re-sizes the pools twice per step and left the workers spinning all the same."""
    global _blas_limit
    if _blas_limit is not None:
        return
    _blas_limit = False
    if os.environ.get('UNIRES_BLAS_THREADS'):
        n = int(os.environ['UNIRES_BLAS_THREADS'])
    if n <= 0:
        return
    try:
        import scipy.linalg  # noqa: F401  (its own OpenBLAS copy loads with it: limit that one too)
        from threadpoolctl import threadpool_limits
        _blas_limit = threadpool_limits(limits=n, user_api='blas')  # kept alive: never restored
    except Exception:
        _blas_limit = False


def light_host(fn):
    """Decorator: ``fn`` is a host section made of tiny matrices - see `cap_threads`, `cap_blas` - applied in batch
    mode (`configure_host(world > 1)` / `batch.init_from_env`) or with UNIRES_LIGHT_HOST=1."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        # (sticky, process-wide: only where ranks share a host - a lone process keeps its pools; the caps save
        # CPU time there, not wall time: tools/rigid_profile.py)
        if batch_mode:
            cap_threads()
            cap_blas()
        return fn(*a, **k)
    return wrapped


class StreamMark:
    """A word of mapped host memory that the device sets when a stream reaches the point of `signal()`
    (``unires_mark_*``, include/unires_hip.h): the host follows its GPU by READING MEMORY.

    Why not events: on this runtime `torch.cuda.synchronize()` and even `Event(blocking=True).synchronize()`
    burn a core for as long as they wait (tools/wait_probe.py -> profiles/r05_wait_probe.txt), and every
    `Event.query()` / stream query on running work makes the runtime submit a marker packet and wakes its
    signal thread, which polls for a while before it sleeps again - a 0.5 ms event poll cost that helper thread
    4.3 ms of CPU per 13.6 ms ADMM iteration (tools/host_profile.py -> profiles/r05_host_profile.txt)."""

    def __init__(self):
        import ctypes
        from . import _lib
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.unires_mark_create(ctypes.byref(self._h)))
        self._val = ctypes.c_uint64(0)
        self._ref = ctypes.byref(self._val)
        self.count = 0  # value of the last signal
        self._last_stream = None

    def signal(self, stream=None):
        """Enqueue "set the word to the next value" on ``stream`` (default: the current one); returns it."""
        from . import _lib
        st = stream if stream is not None else torch.cuda.current_stream()
        self.count += 1
        _lib.check(self._lib.unires_mark_signal(self._h, self.count, st.cuda_stream))
        self._last_stream = st
        return self.count

    def reached(self, value):
        self._lib.unires_mark_read(self._h, self._ref)
        return self._val.value >= value

    def wait(self, value, dt=1e-4, stream=None):
        """Sleep until the device has set the word to ``value`` or beyond (``dt`` seconds between looks).
        A wait that lasts longer than a second starts asking the stream itself twice a second (that call costs
        the runtime something, hence not earlier): a faulted stream raises there, and a stream that has finished
        without the mark being set - which cannot happen to a healthy one - raises here instead of hanging."""
        t0 = t_ask = None
        while not self.reached(value):
            time.sleep(dt)
            if t0 is None:
                t0 = t_ask = time.monotonic()
                continue
            now = time.monotonic()
            if now - t0 > 1.0 and now - t_ask > 0.5:
                t_ask = now
                st = stream if stream is not None else self._last_stream
                if st is not None and st.query() and not self.reached(value):
                    raise RuntimeError('unires_amd: the stream finished without reaching its mark (value %d)' % value)

    def __del__(self):
        try:
            if self._h:
                self._lib.unires_mark_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Pacer:
    """Keeps the host at most ``depth`` steps ahead of the device, SLEEPING while it waits.

    Without it a loop of enqueue-only steps (the ADMM iterations: no read-back in between) fills the hardware
    queue and every further launch call spins inside the runtime until there is room."""

    def __init__(self, depth=2):
        self.depth = max(1, int(depth))
        self._mark = None
        self._stream = None
        self._pending = []

    def step(self, stream=None):
        """Call once per enqueued step, after its last launch."""
        if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return
        st = stream if stream is not None else torch.cuda.current_stream()
        if self._mark is None or self._stream != st:
            # (a mark counts along ONE stream; a loop that moves to another stream starts a new one)
            self.drain()
            self._mark, self._stream = StreamMark(), st
        if len(self._pending) >= self.depth:
            self._mark.wait(self._pending.pop(0), 5e-4)  # (a step `depth` steps old: no hurry)
        self._pending.append(self._mark.signal(st))

    def drain(self):
        while self._pending:
            self._mark.wait(self._pending.pop(0))


_wait_marks = {}


def wait_blocking(device=None):
    """Wait for everything enqueued on the current stream WITHOUT spinning: what `tensor.cpu()` /
    `torch.cuda.synchronize()` do by polling flat out.  (The caller still synchronises / copies afterwards -
    that is what orders the results for the host; after this it no longer waits.)"""
    if not torch.cuda.is_available():
        return
    st = torch.cuda.current_stream(device)
    if torch.cuda.is_current_stream_capturing():
        return
    key = (st.device.index, st.cuda_stream)
    with torch.cuda.device(st.device):
        mark = _wait_marks.get(key)
        if mark is None:
            if len(_wait_marks) > 64:  # (streams come and go: do not collect marks for ever)
                _wait_marks.clear()
            mark = _wait_marks[key] = StreamMark()
        mark.wait(mark.signal(st), 5e-5)
