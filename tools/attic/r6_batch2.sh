#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== tests (GEN rows dealt over waves)"
python -m pytest tests/test_gpu_midsize.py tests/test_gpu_sizes.py tests/test_gpu_orient.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_path.py tests/test_gpu_ops.py -m gpu -x -q -k "profile or thick or axes or xyz or pull" 2>&1 | tail -3
echo "== kernels xyz / demo"
for wl in cfg3_256c3_thick6xyz demo_181c3_thick4xyz; do bash tools/r6_k.sh $wl "0 1"; done
echo "== repeatability of the streams figure (cfg3 z), 3 runs each"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('$1 it/s %6.0f (serial %.0f)' % (d['value'], d['value_channels_serial'] or 0))"; }
for s in 64 28 24; do for rep in 1 2 3; do UNIRES_SHARE_S2=$s python bench.py --no-cpu-baseline --no-variants --admm-iters 1 2>/dev/null | line S2=$s; done; done
echo "== splat ablation (abl build)"
export UNIRES_LIB=$PWD/build/ab/abl.so
for dbg in 0 1 2 4 8; do
  echo "-- UNIRES_S2_DBG=$dbg"
  UNIRES_S2_DBG=$dbg WL=cfg3_256c3_thick6z CH=1 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "k_splat2<\|k_pull_conv2"
done
