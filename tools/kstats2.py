# first lines of a rocprofv3 --stats kernel summary: name, calls, average duration
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), '%8.2f us' % (float(r['AverageNs']) / 1e3), r['Percentage'])
