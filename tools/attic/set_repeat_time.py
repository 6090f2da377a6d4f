# What one unires_plan_set_repeat costs after a rigid change (wall, device idle at entry): the schedule and window-plan
# rebuild that follows every rigid Gauss-Newton update.   WL=cfg3_256c3_thick6z python tools/set_repeat_time.py
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import unires_amd as U
from unires_amd._project import _channel_plan
from unires_amd._rigid import _expm
dev = torch.device('cuda:0')
name = os.environ.get('WL', 'cfg3_256c3_thick6z')
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS[name], dev, seed=1234)
plan = _channel_plan(x[0], y[0], sett.method, sett.do_proj)
xn = x[0][0]
R0 = xn.po.rigid.clone()
ts = []
for k in range(12):
    q = torch.tensor([0.01 * (k + 1), -0.02, 0.015, 1e-3 * k, -5e-4, 2e-4], dtype=torch.float64)
    xn.po.rigid = _expm(q, U.affine_basis('SE')) @ R0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.set_repeat(0, xn.po, xn.tau)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({'workload': name, 'set_repeat_ms': [round(t, 3) for t in ts], 'median_ms': sorted(ts)[len(ts) // 2]}))
