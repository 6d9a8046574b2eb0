"""The largest idle gaps of the busiest queue over the last two steady-state steps of a rocprofv3 kernel trace (argv[1] = dir):
gap length, the kernel before and the kernel after — where the main stream waits for another stream or for the host."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in csv.DictReader(open(f))]
rows.sort()
gru = [s for s, e, n, q in rows if 'gru_split' in n or 'gru_pair' in n]
t0, t1 = gru[-5], gru[-1]
byq = collections.defaultdict(list)
for s, e, n, q in rows:
    if t0 <= s < t1:
        byq[q].append((s, e, n))
q, ks = max(byq.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))
gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '')[:48]
tot = sum(g for g, _, _ in gaps)
print('queue %s: %d kernels in two steps, idle %.2f ms per step' % (q, len(ks), tot / 2e6))
hist = collections.Counter()
for g, _, _ in gaps:
    hist['<3us' if g < 3e3 else '<6us' if g < 6e3 else '<12us' if g < 12e3 else '<30us' if g < 30e3 else '>=30us'] += g
print('idle by gap length (ms per step):', {k: round(v / 2e6, 3) for k, v in hist.items()})
for g, a, b in sorted(gaps, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print('%8.1f us   %-48s -> %s' % (g / 1e3, short(a), short(b)))
