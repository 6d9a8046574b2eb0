"""per-step kernel time by family over the last two steady-state steps of a kernel trace"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
gru = [s for s, e, n in rows if 'gru_split' in n or 'gru_pair' in n]
t0, t1 = gru[-5], gru[-1]
win = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
fam = collections.Counter(); cnt = collections.Counter()
def key(n):
    for k in ('attn_modes', 'attn_fused', 'attn_fwd', 'fourier_kernel', 'attn_post', 'ln_linear', 'gru', 'mlp_heads', 'radius', 'rocprim', 'copyBuffer', 'at::native'):
        if k in n: return k
    return n[:40]
for s, e, n in win:
    fam[key(n)] += e - s; cnt[key(n)] += 1
for k, v in fam.most_common(20):
    print('%-40s %8.2f ms/step  %5d launches/step' % (k, v / 2e6, cnt[k] // 2))
