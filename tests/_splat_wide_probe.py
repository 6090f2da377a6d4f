"""Prints sha256 digests of At v and AtA p for thick slices along x, y and z on a 64 x 32 x 60 volume
(128 splat tiles = 32 four-wave workgroups), and of the single-pass A^T A of the denoising regime.  test_gpu_ops.py
runs it as is, with the persistent grids capped by the runtime's occupancy answer (UNIRES_SPLAT2_RESIDENT=1) and with
the plans told about neighbours (PROBE_CONCURRENCY=3 -> unires_plan_set_concurrency; UNIRES_SHARE_S2 / _F1 = 1 make
the caps that follow 16 workgroups, below this volume's grids): the same tiles, walked by fewer waves."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unires_amd as U  # noqa: E402
from tests.helpers import rigid_matrix  # noqa: E402
from unires_amd._plan import ChannelPlan  # noqa: E402

dev = torch.device('cuda:0')
dim_y = (64, 32, 60)
mat_y = torch.eye(4, dtype=torch.float64)
rigid = rigid_matrix([0.4, -0.3, 0.2], [0.03, -0.02, 0.04])
for axis in range(3):
    sc = [1.0, 1.0, 1.0, 1.0]
    sc[axis] = 4.0
    mat_x = mat_y @ torch.diag(torch.tensor(sc, dtype=torch.float64))
    dim_x = tuple(d // 4 if a == axis else d for a, d in enumerate(dim_y))
    po = U._proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, device=dev, scl=0.1)
    torch.manual_seed(3 + axis)
    p = (torch.rand(dim_y) + 0.5).to(dev)
    v = (torch.rand(dim_x) + 0.5).to(dev)
    # plan-level operators: the schedule-driven splat of the hot path (U._proj_apply runs the op-level kernels)
    plan = ChannelPlan(dim_y, (1.0, 1.0, 1.0), [(po, 1.0)], 'super-resolution', True, device=dev)
    plan.set_concurrency(int(os.environ.get('PROBE_CONCURRENCY', '1')))
    for op, arg in (('At', v), ('AtA', p)):
        out = plan.proj_apply(0, op, arg)
        print(op, axis, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())
    q = plan.matvec(p, 0.7, 0.05)
    print('AtA', 3 + axis, hashlib.sha256(q.cpu().numpy().tobytes()).hexdigest())

# denoising regime: pull + push + stencil in one pass (k_ata1) on a rotated 1 mm observation of the same volume
po = U._proj_info(dim_y, mat_y, dim_y, mat_y, rigid=rigid, device=dev)
plan = ChannelPlan(dim_y, (1.0, 1.0, 1.0), [(po, 1.0)], 'denoising', True, device=dev)
plan.set_concurrency(int(os.environ.get('PROBE_CONCURRENCY', '1')))
torch.manual_seed(11)
p = (torch.rand(dim_y) + 0.5).to(dev)
print('AtA', 6, hashlib.sha256(plan.matvec(p, 0.7, 0.05).cpu().numpy().tobytes()).hexdigest())
