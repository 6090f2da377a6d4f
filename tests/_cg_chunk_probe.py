"""Prints, for three seeded y-update problems (thick-slice, pull-only, A = I), every channel's realised CG
iteration count, the sha256 of its iterate and its objective trace under the reference-default stopping
rule (tolerance 1e-3, 'max_gain') and under the residual rule.  tests/test_gpu_cg.py runs it with
UNIRES_CG_CHUNK = 0 (the whole solve enqueued at once, rounds 1-3), 1, 2 (default), 3 and 7 and with the
hipGraph replay off: the chunked enqueue feeds the same kernels in the same order, so nothing may
change by a bit."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_problem, run_gpu_update_y  # noqa: E402

CASES = [dict(dim_y=(20, 18, 16), n_channels=3, thick=4, regime='sr', scl=0.1),
         dict(dim_y=(15, 13, 11), n_channels=2, regime='dn', rot=0.1, trans=1.5),
         dict(dim_y=(18, 17, 13), n_channels=1, regime='id')]
for i, kw in enumerate(CASES):
    prob = make_problem(seed=40 + i, **kw)
    for stop, tol in (('max_gain', 1e-3), ('e', 1e-2), ('max_gain_recurred', 1e-3)):
        for rep in range(2):  # the second call replays the captured graphs
            y, info = run_gpu_update_y(prob, 'cuda:0', max_iter=20, tol=tol, stop=stop)
        for c in range(len(y)):
            print('CG', i, stop, c, info[c][0], hashlib.sha256(y[c].cpu().numpy().tobytes()).hexdigest(),
                  ' '.join('%.17g' % v for v in info[c][1]))
