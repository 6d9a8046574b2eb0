"""Per-launch HBM traffic of the fused attention kernel from two rocprofv3 --pmc passes.
usage: pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass>"""
import csv, glob, sys, json


def load(d, counter):
    f = (glob.glob(d + '/*/*counter_collection.csv') + glob.glob(d + '/*counter_collection.csv'))[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    return rows


def by_kernel(rows, pat):
    return [float(r['Counter_Value']) for r in rows if pat in r['Kernel_Name']]


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
cal_bytes = (4 << 20) * 512.0
cal_f = by_kernel(fetch, 'row_normalize')[-3:]
cal_w = by_kernel(write, 'row_normalize')[-3:]
# counter unit is KiB per dispatch; calibration = known bytes / reported bytes in this access width (8 B per lane)
kf = cal_bytes / (sum(cal_f) / len(cal_f) * 1024.0)
kw = cal_bytes / (sum(cal_w) / len(cal_w) * 1024.0)
def attn(rows):
    return [float(r['Counter_Value']) for r in rows if 'attn_fused_kernel' in r['Kernel_Name'] or 'attn_modes' in r['Kernel_Name'] or 'attn_kv_kernel' in r['Kernel_Name']]


af, aw = attn(fetch), attn(write)
n = len(af) // 2            # two predict calls; take the second (warm) one
af, aw = af[n:], aw[n:]
out = {'calibration': {'known_bytes': cal_bytes, 'fetch_factor': kf, 'write_factor': kw,
                       'raw_fetch_KiB': cal_f, 'raw_write_KiB': cal_w},
       'attn_fused': {'launches': len(af), 'fetch_bytes_per_launch': sum(af) * 1024.0 * kf / len(af),
                      'write_bytes_per_launch': sum(aw) * 1024.0 * kw / len(aw)}}
out['attn_fused']['traffic_bytes_per_launch'] = out['attn_fused']['fetch_bytes_per_launch'] + out['attn_fused']['write_bytes_per_launch']
print(json.dumps(out, indent=1))
