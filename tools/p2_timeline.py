# per-workgroup timeline of k_pull_conv2 (-DUNIRES_P2_PROF build, UNIRES_P2_PROF_OUT=<file>)
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.float64, comments='#')
print(open(sys.argv[1]).readline().strip())
t0 = a[:, 0].min()
start, staged, end = (a[:, 0] - t0) / 100, (a[:, 1] - t0) / 100, (a[:, 2] - t0) / 100
hw = a[:, 4].astype(np.int64)
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4)   # HW_ID: CU_ID 11:8, SE_ID 15:13 (XCC not in this register)
print('workgroups %d  start: med %.1f max %.1f us  end: med %.1f max %.1f us' % (len(a), np.median(start), start.max(), np.median(end), end.max()))
print('per workgroup: staging med %.2f us (p90 %.2f)  sampling + conv med %.2f us (p90 %.2f)  life med %.2f' % (
    np.median(staged - start), np.percentile(staged - start, 90), np.median(end - staged), np.percentile(end - staged, 90), np.median(end - start)))
# concurrency over time
ts = np.linspace(0, end.max(), 12)[1:-1]
print('workgroups in flight at t =', ', '.join('%.0f us: %d' % (t, ((start <= t) & (end > t)).sum()) for t in ts))
print('sum of workgroup lives / (kernel time x 256 CUs) = %.2f workgroups per CU on average' % ((end - start).sum() / (end.max() * 256)))
