#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_guard.py tests/test_gpu_cg.py tests/test_gpu_sizes.py tests/test_golden.py tests/test_gpu_midsize.py -m gpu -x -q 2>&1 | tail -4
python tools/r6_gains.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
python bench.py --no-cpu-baseline --admm-iters 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); v=d['variants']['cg_tol1e-3_max_gain']; print(v['cg_iters_realised'], v['ms_per_step'], v['ms_per_realised_iteration'], d['subjects_per_sec_tol1e-3'], d['subjects_per_sec'], d['value'])"
