# per-wave timeline of k_splat2 (-DUNIRES_S2_PROF build, UNIRES_S2_PROF_OUT=<file>): where a wave's life goes
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.float64)
t0 = a[:, 0].min()
us = lambda t: (t - t0) / 100.0
start, end, nt = us(a[:, 0]), us(a[:, 1]), a[:, 2].astype(int)
print('waves %d  start: min %.1f med %.1f max %.1f us   end: min %.1f med %.1f p90 %.1f max %.1f us' % (
    len(a), start.min(), np.median(start), start.max(), end.min(), np.median(end), np.percentile(end, 90), end.max()))
print('tiles per wave: ', np.bincount(nt))
for k in sorted(set(nt)):
    m = nt == k
    print('  %d tiles: %5d waves, life med %.1f us, end med %.1f max %.1f' % (k, m.sum(), np.median(end[m] - start[m]), np.median(end[m]), end[m].max()))
st, ep, ni = [], [], []
for w in range(len(a)):
    prev = a[w, 0]
    for i in range(min(nt[w], 9)):
        ts, te, n = a[w, 3 + 3 * i], a[w, 4 + 3 * i], a[w, 5 + 3 * i]
        st.append((ts - prev) / 100.0); ep.append((te - ts) / 100.0); ni.append(n)
        prev = te
st, ep, ni = np.array(st), np.array(ep), np.array(ni)
print('tiles %d  instr/tile med %.0f  stream: med %.2f mean %.2f us (%.3f us / instr)   epilogue: med %.2f mean %.2f p90 %.2f us' % (
    len(st), np.median(ni), np.median(st), st.mean(), st.sum() / max(ni.sum(), 1), np.median(ep), ep.mean(), np.percentile(ep, 90)))
# by tile ordinal
for i in range(6):
    s_i = [(a[w, 3 + 3 * i] - (a[w, 0] if i == 0 else a[w, 4 + 3 * (i - 1)])) / 100.0 for w in range(len(a)) if nt[w] > i]
    e_i = [(a[w, 4 + 3 * i] - a[w, 3 + 3 * i]) / 100.0 for w in range(len(a)) if nt[w] > i]
    if s_i: print('  tile #%d: n %5d  stream med %.2f  epilogue med %.2f us' % (i, len(s_i), np.median(s_i), np.median(e_i)))
