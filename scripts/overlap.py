"""Who runs beside whom: for the last two steady-state steps of a rocprofv3 kernel trace (argv[1] = output dir), per kernel
family its launches, summed duration and the share of that time during which a kernel of ANOTHER queue was running too
(by the family of the co-runner)."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in csv.DictReader(open(f))]
rows.sort()
FAM = ('attn_modes', 'attn_kv', 'attn_fused', 'attn_fwd', 'fourier_pair', 'fourier_kernel', 'attn_post', 'ln_linear', 'gru', 'mlp_heads',
       'radius', 'rocprim', 'at::native')
def fam(n):
    for k in FAM:
        if k in n:
            return k
    return 'other'
gru = [s for s, e, n, q in rows if 'gru_split' in n or 'gru_pair' in n]
t0, t1 = gru[-5], gru[-1]
win = [(s, e, fam(n), q) for s, e, n, q in rows if t0 <= s < t1]
tot = collections.Counter(); cnt = collections.Counter(); co = collections.defaultdict(collections.Counter)
for i, (s, e, k, q) in enumerate(win):
    tot[k] += e - s; cnt[k] += 1
    j = i - 1
    while j >= 0 and win[j][0] > s - 20_000_000:
        s2, e2, k2, q2 = win[j]
        if q2 != q and e2 > s:
            co[k][k2] += min(e, e2) - s
        j -= 1
    j = i + 1
    while j < len(win) and win[j][0] < e:
        s2, e2, k2, q2 = win[j]
        if q2 != q:
            co[k][k2] += min(e, e2) - s2
        j += 1
print('queues:', sorted(set(q for _, _, _, q in win)))
for k, v in tot.most_common():
    share = ', '.join('%s %.0f%%' % (k2, 100.0 * c / v) for k2, c in co[k].most_common(4))
    print('%-16s %4d launches/step %7.2f ms/step avg %7.1f us | beside: %s' % (k, cnt[k] // 2, v / 2e6, v / 1e3 / cnt[k], share))
