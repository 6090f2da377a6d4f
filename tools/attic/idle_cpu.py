# process CPU time (all threads) while the main thread (a) sleeps, (b) waits for GPU work by query + sleep - after the
# same initialisation as the fit tools: is there a background burn in the runtime's helper threads?
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
dev = torch.device('cuda:0')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS['demo_181c3_thick4xyz'], dev, seed=1234)
torch.cuda.synchronize()
out = {}
c0, t0 = time.process_time(), time.perf_counter(); time.sleep(1.0)
out['sleep_1s'] = {'cpu_ms': (time.process_time() - c0) * 1e3, 'wall_ms': (time.perf_counter() - t0) * 1e3}
a = torch.rand(4096, 4096, device=dev)
c0, t0 = time.process_time(), time.perf_counter()
for _ in range(200):
    b = a @ a
    ev = torch.cuda.Event(); ev.record()
    while not ev.query():
        time.sleep(1e-4)
out['200_waits'] = {'cpu_ms': (time.process_time() - c0) * 1e3, 'wall_ms': (time.perf_counter() - t0) * 1e3}
c0, t0 = time.process_time(), time.perf_counter()
for _ in range(200):
    v = float(a[0, 0].cpu())
out['200_scalar_readbacks'] = {'cpu_ms': (time.process_time() - c0) * 1e3, 'wall_ms': (time.perf_counter() - t0) * 1e3}
h = torch.empty(18432 * 2, dtype=torch.int32)
d = torch.zeros(18432 * 2, dtype=torch.int32, device=dev)
c0, t0 = time.process_time(), time.perf_counter()
for _ in range(200):
    h.copy_(d)
out['200_pageable_150KB_d2h'] = {'cpu_ms': (time.process_time() - c0) * 1e3, 'wall_ms': (time.perf_counter() - t0) * 1e3}
print(json.dumps(out))
