"""Per-launch view of a rocprofv3 kernel trace: kernels whose name contains argv[2], grouped by grid size."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
pat = sys.argv[2]
g = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        key = (r['Kernel_Name'][:60], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Workgroup_Size_X']))
        g.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (name, blocks, wg), d in g.items():
    print('%-60s blocks %7d x %4d  calls %4d  avg %9.1f us  total %8.2f ms' % (name, blocks, wg, len(d), sum(d) / len(d), sum(d) / 1e3))
