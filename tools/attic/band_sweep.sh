cd $GRAFT_REPO_ROOT
for wl in cfg4_384c4_iso2 cfg4_384c4_iso2_gauss; do
for b in 0 1 2 8 1000; do
  echo "== $wl band $b"
  UNIRES_P2_BAND=$b CH=0 WL=$wl bash tools/traffic2.sh x -- python $GRAFT_REPO_ROOT/tools/pmc5.py | grep k_pull_conv2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('fetch MB %.1f write MB %.1f' % (2*d['FETCH_SIZE']/1024, d['WRITE_SIZE']/1024))"
  UNIRES_P2_BAND=$b WL=$wl CH=0 bash tools/prof.sh tools/pmc5.py 2>&1 | grep "k_pull_conv2"
done; done
