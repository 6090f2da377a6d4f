import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'unires' in n:
        agg[n[:58]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n,v in agg.items():
    v2=sorted(v)
    print(n.ljust(58), 'n=%4d'%len(v), 'med %8.1f  min %8.1f  max %8.1f us' % (v2[len(v2)//2], v2[0], v2[-1]))
