"""Largest idle gaps of the GPU inside the last two steady-state steps of a rocprofv3 kernel trace (argv[1] = dir):
for every gap the kernels that end before / start after it, and the gap histogram."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r.get('Stream_Id', '')) for r in csv.DictReader(open(f))]
rows.sort()
gru = [s for s, e, n, q, st in rows if 'gru_split' in n or 'gru_pair' in n]
t0, t1 = gru[-5], gru[-1]
win = [r for r in rows if t0 <= r[0] < t1]
gaps = []
cur_e, last = None, None
for r in win:
    s, e = r[0], r[1]
    if cur_e is not None and s > cur_e:
        gaps.append((s - cur_e, last, r))
    if cur_e is None or e > cur_e:
        cur_e, last = e, r
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '')
    return n[:60]
tot = sum(g[0] for g in gaps)
print('gaps: %d, total %.2f ms per step' % (len(gaps), tot / 2e6))
for lo, hi in ((0, 5), (5, 20), (20, 100), (100, 1000), (1000, 1e9)):
    sel = [g[0] for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
    print('  %5g-%5g us: %5d gaps, %.2f ms per step' % (lo, hi, len(sel), sum(sel) / 2e6))
for g, a, b in sorted(gaps, key=lambda x: -x[0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print('%8.1f us at +%7.2f ms  after %-60s [q %s] before %-60s [q %s]' % (g / 1e3, (b[0] - t0) / 1e6, short(a[2]), a[3], short(b[2]), b[3]))
