import csv, sys, glob
f = (glob.glob(sys.argv[1] + '/*/*kernel_stats.csv') + glob.glob(sys.argv[1] + '/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6, 'n kernels', len(rows))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print('%-80s calls %6s total %8.2f ms avg %8.1f us  %5.1f%%' % (r['Name'][:80], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['Percentage'])))
