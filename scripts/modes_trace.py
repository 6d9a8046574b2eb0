"""Per-launch durations of attn_modes / split_rows kernels from a rocprofv3 kernel trace (argv[1] = trace csv)."""
import csv, collections, sys
g = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "attn_modes" in name or "split_rows" in name:
        key = (name[:52], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
        g.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in g.items():
    d.sort()
    print(k, "calls", len(d), "avg us %.1f" % (sum(d) / len(d)), "min %.1f med %.1f max %.1f" % (d[0], d[len(d) // 2], d[-1]))
