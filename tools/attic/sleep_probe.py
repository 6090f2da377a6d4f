import time, torch, json
ev = torch.cuda.Event(); a = torch.rand(8192, 8192, device='cuda'); torch.cuda.synchronize()
out = {}
for dt in (5e-5, 1e-4, 2e-4, 3e-4, 5e-4, 1e-3):
    for _ in range(20): b = a @ a
    ev = torch.cuda.Event(); ev.record()
    c0, t0, n = time.process_time(), time.perf_counter(), 0
    while not ev.query():
        time.sleep(dt); n += 1
    out[str(dt)] = {'polls': n, 'cpu_ms': round((time.process_time() - c0) * 1e3, 2), 'wall_ms': round((time.perf_counter() - t0) * 1e3, 1)}
print(json.dumps(out))
