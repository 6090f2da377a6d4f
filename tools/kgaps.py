"""Gaps between consecutive kernels of a rocprofv3 kernel trace (csv): how much of a launch-bound solve is spent
between kernels.  usage: kgaps.py k_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
busy = gap = 0
gaps = []
for a, b in zip(rows, rows[1:]):
    busy += int(a['End_Timestamp']) - int(a['Start_Timestamp'])
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if 0 <= g < 50000:  # (longer: host pauses between solves)
        gap += g
        gaps.append(g)
gaps.sort()
n = len(gaps)
print('kernels %d  busy %.2f ms  short gaps %.2f ms  (median %.2f us, p90 %.2f us)' %
      (len(rows), busy / 1e6, gap / 1e6, gaps[n // 2] / 1e3, gaps[int(0.9 * n)] / 1e3))
