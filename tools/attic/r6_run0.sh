cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-variants --admm-iters 3"
pr() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); r=d['roofline']; print('$1', 'it/s %.0f ms/step %.3f mv %.1f cold %.1f bych %s subj/s %.3f tol %.3f' % (d['value'], d['ms_per_step'], r['us_per_launch'], r['us_per_launch_cold'], ['%.1f'%v for v in r['us_per_launch_by_channel']], d['subjects_per_sec'], d['subjects_per_sec_tol1e-3']))"; }
$B 2>/dev/null | pr base
$B --channel-streams 2>/dev/null | pr streams1024
UNIRES_SPLAT2_BLOCKS=448 $B --channel-streams 2>/dev/null | pr streams448
UNIRES_SPLAT2_BLOCKS=512 $B --channel-streams 2>/dev/null | pr streams512
