"""Gaps between consecutive kernels of one queue (default: the busiest) in the last two steady-state steps of a rocprofv3
kernel trace (argv[1] = dir): percentiles in us and the sum per step — small gaps are the device's own kernel-to-kernel
latency, large ones the host (under the profiler) or a wait for another stream."""
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
for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    gaps = sorted(max(0, ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1))
    n = len(gaps)
    small = [g for g in gaps if g < 30]
    print('queue %s: %d kernels/step, gaps p10 %.1f p50 %.1f p90 %.1f us, sum %.2f ms/step, gaps < 30 us: %d/step summing %.2f ms/step'
          % (q, (n + 1) // 2, gaps[n // 10], gaps[n // 2], gaps[9 * n // 10], sum(gaps) / 2e3, len(small) // 2, sum(small) / 2e3))
