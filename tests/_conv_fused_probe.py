"""Prints the sha256 of A^T A p, A p and A^T x of every channel of four ratio-2 problems with many-tap profiles
(the `_v4` cases of tests/test_gpu_path.py: sizes where the 16-byte separable passes apply).
tests/test_gpu_path.py::test_one_kernel_conv_passes_are_bit_identical_to_the_separate_ones runs it with the
one-kernel forms on (default) and with UNIRES_CONV_YX=0 / UNIRES_CONV_DOWNUP=0 / UNIRES_UPYZ_LDS=0 (the separate
marching passes they replace): same products in the same order, so nothing may change by a bit."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_problem, gpu_structs  # noqa: E402
import unires_amd as U  # noqa: E402

CASES = [dict(dim_y=(20, 24, 32), n_channels=1, thick=2, regime='sr', iso=True, prof_ip=2, scl=0.05),
         dict(dim_y=(20, 24, 32), n_channels=1, thick=2, regime='sr', iso=True, prof_ip=2, prof_tp=1),
         dict(dim_y=(24, 20, 32), n_channels=1, thick=2, regime='sr', iso=True, prof_ip=2, prof_tp=2, scl=0.05),
         dict(dim_y=(24, 20, 32), n_channels=2, thick=2, regime='sr', iso=True, scl=0.1),
         dict(dim_y=(24, 24, 32), n_channels=3, thick=2, regime='sr', iso=True, prof_ip=2, scl=0.05,
              orient=[((0, 1, 2), (0, 0, 0)), ((1, 0, 2), (0, 1, 0)), ((2, 1, 0), (0, 0, 1))])]
dev = torch.device('cuda:0')
for i, kw in enumerate(CASES):
    prob = make_problem(seed=60 + i, **kw)
    xg, yg, sett = gpu_structs(prob, dev)
    torch.manual_seed(9)
    for c in range(len(xg)):
        p = torch.rand(prob['dim_y'], device=dev) * 100
        xv = torch.rand(tuple(xg[c][0].po.dim_x), device=dev)
        outs = [U._proj('AtA', p, xg[c], yg[c], method=prob['method'], do=True, rho=torch.tensor(prob['rho']),
                        vx_y=torch.ones(3)),
                U._proj('A', p, xg[c], yg[c], method=prob['method'], do=True, n=0),
                U._proj('At', xv, xg[c], yg[c], method=prob['method'], do=True, n=0)]
        torch.cuda.synchronize()
        print('CONV', i, c, ' '.join(hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16] for o in outs))
