"""Largest non-qcn kernels (torch / rocPRIM / copies) of the last two steady-state steps of a rocprofv3 kernel trace
(argv[1] = output dir): per (kernel name, queue) launches and time per step."""
import csv, sys, glob, collections, re
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'), r['Grid_Size_X']) for r in csv.DictReader(open(f))]
rows.sort()
gru = [s for s, e, n, q, g in rows if 'gru_split' in n or 'gru_pair' in n]
t0, t1 = gru[-5], gru[-1]
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, q, g in rows:
    if not (t0 <= s < t1) or 'qcn' in n or '(anonymous namespace)' in n and 'at::' not in n:
        continue
    short = re.sub(r'<.*', '', n)[:70]
    m = re.search(r'(\w+_kernel_cuda|\w+Functor|\w+_functor|index\w*|gather\w*|CatArray\w*|copy\w*)', n)
    key = (short + ' ' + (m.group(1) if m else ''), q)
    tot[key] += e - s; cnt[key] += 1
for k, v in tot.most_common(40):
    print('%-110s q%s %5d/step %7.3f ms/step avg %6.1f us' % (k[0][:110], k[1], cnt[k] // 2, v / 2e6, v / 1e3 / cnt[k]))
