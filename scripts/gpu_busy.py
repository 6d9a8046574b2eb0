"""GPU busy fraction over whole steady-state steps of a rocprofv3 kernel trace (argv[1] = kernel_trace.csv): the window
spans the last two bench steps (delimited by the gru_split_kernel launches, two per step)."""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
gru = [s for s, e, n in rows if 'gru_split' in n or 'gru_pair' in n]
t0, t1 = gru[-5], gru[-1]
win = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
busy = sum(e - s for s, e, n in win)
ours = sum(e - s for s, e, n in win if 'anonymous namespace' in n and 'at::' not in n)
small = [(s, e, n) for s, e, n in win if not ('anonymous namespace' in n and 'at::' not in n)]
# union of busy intervals (kernels may overlap)
cur_s, cur_e, union = None, None, 0
for s, e, n in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print('window %.2f ms (2 steps): kernels %d, sum of durations %.2f ms, union busy %.2f ms (%.1f%%), idle %.2f ms per step'
      % ((t1 - t0) / 1e6, len(win), busy / 1e6, union / 1e6, 100.0 * union / (t1 - t0), (t1 - t0 - union) / 2e6))
print('hand-written kernels %.2f ms per step, other (torch / rocprim / copies) %.2f ms per step in %d launches per step'
      % (ours / 2e6, (busy - ours) / 2e6, len(small) // 2))
