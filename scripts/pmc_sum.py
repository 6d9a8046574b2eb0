"""Sum of every counter over the dispatches of kernels whose name contains argv[2], from a rocprofv3 --pmc csv dir (argv[1])."""
import csv, glob, sys, collections
f = (glob.glob(sys.argv[1] + '/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*/*counter_collection.csv'))[0]
tot, n = collections.OrderedDict(), collections.Counter()
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r['Kernel_Name']:
        tot[r['Counter_Name']] = tot.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        n[r['Counter_Name']] += 1
for k, v in tot.items():
    print('%-32s %16.0f  (%d dispatches, %.0f per dispatch)' % (k, v, n[k], v / n[k]))
