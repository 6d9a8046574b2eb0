"""Kernel time per queue (stream) and family over the last two steady-state steps of a rocprofv3 kernel trace (argv[1] = dir)."""
import csv, sys, glob, collections, re
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in csv.DictReader(open(f))]
rows.sort()
gru = [s for s, e, n, q in rows if 'gru_split' in n or 'gru_pair' in n]
t0, t1 = gru[-5], gru[-1]
FAM = ('attn_modes', 'attn_kv', 'attn_fused', 'attn_fwd', 'fourier_pair', 'fourier_kernel', 'attn_post', 'ln_linear', 'gru', 'mlp_heads',
       'radius', 'rocprim', 'at::native', 'mask_pairs', 'rel_geometry', 'csr_', 'coalesce', 'cumsum', 'copyBuffer', 'fillBuffer',
       'motion_metrics', 'to_agent_frame', 'split_rows', 'row_normalize')
def fam(n):
    for k in FAM:
        if k in n:
            return k
    return re.sub(r'<.*', '', n)[:40]
tot = collections.defaultdict(collections.Counter); cnt = collections.defaultdict(collections.Counter)
for s, e, n, q in rows:
    if t0 <= s < t1:
        tot[q][fam(n)] += e - s; cnt[q][fam(n)] += 1
for q in sorted(tot):
    print('queue %s: %.2f ms/step in %d launches/step' % (q, sum(tot[q].values()) / 2e6, sum(cnt[q].values()) // 2))
    for k, v in tot[q].most_common(14):
        print('    %-28s %7.2f ms/step %4d launches/step' % (k, v / 2e6, cnt[q][k] // 2))
