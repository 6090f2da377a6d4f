# per-wave timeline of k_splat2 (-DUNIRES_S2_PROF build, UNIRES_S2_PROF_OUT=<file>): where a wave's life goes
# per wave: [0] start, [1] end, [2] tiles, then per tile (first five): header arrived, stream end, p window arrived,
# tile end (stores drained), instructions  (100 MHz ticks)
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.float64)
t0 = a[:, 0].min()
us = lambda t: (t - t0) / 100.0
start, end, nt = us(a[:, 0]), us(a[:, 1]), a[:, 2].astype(int)
print('waves %d  start: min %.1f med %.1f max %.1f us   end: min %.1f med %.1f p90 %.1f max %.1f us' % (
    len(a), start.min(), np.median(start), start.max(), end.min(), np.median(end), np.percentile(end, 90), end.max()))
print('tiles per wave: ', np.bincount(nt))
ph = {k: [] for k in ('header', 'stream', 'pload', 'arith', 'ninstr')}
for w in range(len(a)):
    prev = a[w, 0]
    for i in range(min(nt[w], 5)):
        th, ts, tp, te, n = a[w, 3 + 5 * i: 8 + 5 * i]
        if tp == 0:  # generic epilogue (no p-window stamp)
            tp = ts
        ph['header'].append((th - prev) / 100.0); ph['stream'].append((ts - th) / 100.0)
        ph['pload'].append((tp - ts) / 100.0); ph['arith'].append((te - tp) / 100.0); ph['ninstr'].append(n)
        prev = te
for k in ('header', 'stream', 'pload', 'arith'):
    v = np.array(ph[k])
    print('  %-7s med %.2f mean %.2f p90 %.2f us' % (k, np.median(v), v.mean(), np.percentile(v, 90)))
ni = np.array(ph['ninstr']); st = np.array(ph['stream'])
print('  tiles %d  instr/tile med %.0f  stream %.3f us / instr;  tile total mean %.2f us' % (
    len(ni), np.median(ni), st.sum() / max(ni.sum(), 1), sum(np.array(ph[k]).mean() for k in ('header', 'stream', 'pload', 'arith'))))
