# total time per kernel name (us) and launch count from a rocprofv3 kernel trace; argv[2] = divide by (e.g. y-updates)
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
t_first, t_last = None, None
for r in rows:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg[n[:70]]; a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print('%-70s n %7.1f  total %9.1f us  mean %7.2f' % (n, a[0] / div, a[1] / div, a[1] / a[0]))
print('sum of kernel durations: %.1f us' % (tot / div))
