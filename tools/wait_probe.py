# How a process should wait for its GPU without burning a core (ROCm runtime): process CPU time vs wall time for a
# 2 s stretch of device work awaited by (a) torch.cuda.synchronize(), (b) a blocking-sync event, (c) polling an
# event with sleeps; optionally after hipSetDeviceFlags(hipDeviceScheduleBlockingSync) (MODE=flags, before the
# context exists).  Per-thread CPU from /proc/self/task.
import ctypes, json, os, sys, time
mode = os.environ.get('MODE', 'plain')
if mode == 'flags':
    hip = ctypes.CDLL('libamdhip64.so')
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(4))  # hipDeviceScheduleBlockingSync
    print('hipSetDeviceFlags ->', rc, file=sys.stderr)
import torch
dev = torch.device('cuda:0')
a = torch.rand(8192, 8192, device=dev)
torch.cuda.synchronize()


def threads_cpu():
    out = {}
    for t in os.listdir('/proc/self/task'):
        try:
            f = open('/proc/self/task/%s/stat' % t).read().rsplit(')', 1)[1].split()
            out[t] = (int(f[11]) + int(f[12])) / os.sysconf('SC_CLK_TCK')
        except Exception:
            pass
    return out


def work():
    for _ in range(40):
        (a @ a)


res = {'mode': mode}
for how in ('synchronize', 'blocking_event', 'poll_sleep'):
    torch.cuda.synchronize()
    c0, t0, th0 = time.process_time(), time.perf_counter(), threads_cpu()
    work()
    t_enq = time.perf_counter() - t0
    if how == 'synchronize':
        torch.cuda.synchronize()
    elif how == 'blocking_event':
        ev = torch.cuda.Event(blocking=True)
        ev.record()
        ev.synchronize()
    else:
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            time.sleep(0.0005)
    wall, cpu = time.perf_counter() - t0, time.process_time() - c0
    th1 = threads_cpu()
    busy = sorted(((th1[k] - th0.get(k, 0.0)) for k in th1), reverse=True)[:3]
    res[how] = {'wall_s': round(wall, 3), 'cpu_s': round(cpu, 3), 'enqueue_s': round(t_enq, 3), 'busiest_threads_s': [round(b, 2) for b in busy]}
print(json.dumps(res))
