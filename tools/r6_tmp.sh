export NMV=100
for b in -1 0 1; do
  echo "== UNIRES_S2_DYN=$b"
  for ch in 0 1; do UNIRES_S2_DYN=$b bash tools/r6_k.sh cfg3_256c3_thick6z $ch | grep splat2; done
done
WL=cfg3_256c3_thick6z,small_96c3_thick3,demo_181c3_thick4xyz python tools/r6_cmp.py build/ab/base_r5.so 2>&1 | tail -18
