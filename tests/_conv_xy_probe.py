"""Prints sha256 digests of A p, At v and AtA p for an isotropic 2 x down-sampling whose x-space z
extent is a multiple of 4 (run by test_gpu_ops.py in two processes: UNIRES_CONV_XY=1 / 0 - the
switch is read once per process)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unires_amd as U  # noqa: E402
from tests.helpers import rigid_matrix  # noqa: E402

dev = torch.device('cuda:0')
dim_y = (22, 18, 24)
mat_y = torch.eye(4, dtype=torch.float64)
mat_x = mat_y @ torch.diag(torch.tensor([2.0, 2.0, 2.0, 1.0], dtype=torch.float64))
dim_x = tuple(d // 2 for d in dim_y)
rigid = rigid_matrix([0.4, -0.3, 0.2], [0.03, -0.02, 0.04])
scl = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
prof = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0: rect (fan-in 2 x 2), 1: triangle (3 x 3)
po = U._proj_info(dim_y, mat_y, dim_x, mat_x, rigid=rigid, device=dev, scl=scl, prof_ip=prof, prof_tp=prof)
torch.manual_seed(3)
p = (torch.rand(dim_y) + 0.5).to(dev)
v = (torch.rand(dim_x) + 0.5).to(dev)
# the plan-level operators (the fused pull / conv / splat kernels of the hot path; U._proj_apply composes
# the op-level kernels and never runs the separable passes)
from unires_amd._plan import ChannelPlan  # noqa: E402
plan = ChannelPlan(dim_y, (1.0, 1.0, 1.0), [(po, 1.0)], 'super-resolution', True, device=dev)
for op, arg in (('A', p), ('At', v), ('AtA', p)):
    out = plan.proj_apply(0, op, arg)
    print(op, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())
