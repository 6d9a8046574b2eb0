"""Per-kernel-family SQ counter summary of a rocprofv3 --pmc pass over scripts/pmc_workload.py (second, warm step only).
usage: sq_summary.py <dir>  -> JSON.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections, csv, glob, json, sys

f = (glob.glob(sys.argv[1] + '/*/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*counter_collection.csv'))[0]
FAMILIES = ('attn_modes_kernel', 'attn_kv_kernel', 'attn_fused_kernel', 'attn_fwd_kernel', 'fourier_pair_kernel', 'fourier_kernel', 'attn_post_kernel', 'ln_linear_kernel',
            'gru_split_kernel', 'gru_pair_kernel', 'mlp_heads_kernel')
rows = collections.defaultdict(lambda: collections.defaultdict(list))       # family -> counter -> per-dispatch values
for r in csv.DictReader(open(f)):
    fam = next((k for k in FAMILIES if k in r['Kernel_Name']), None)
    if fam:
        rows[fam][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for fam, ctr in rows.items():
    half = {k: v[len(v) // 2:] for k, v in ctr.items()}                      # two predict calls: keep the warm one
    s = {k: sum(v) for k, v in half.items()}
    n = len(next(iter(half.values())))
    wave = s.get('SQ_WAVE_CYCLES', 0.0)
    gui = s.get('GRBM_GUI_ACTIVE', 0.0) / 8.0                                # cycles of wall time, summed over dispatches
    d = {'dispatches': n, 'raw': s}
    if wave:
        d['wave_cycles_parked_on_waitcnt_or_barrier'] = s.get('SQ_WAIT_ANY', 0.0) / wave
        d['wave_cycles_issue_stalled'] = s.get('SQ_WAIT_INST_ANY', 0.0) / wave
        d['wave_cycles_issuing'] = s.get('SQ_ACTIVE_INST_ANY', 0.0) / wave
        d['wave_cycles_issuing_valu'] = s.get('SQ_ACTIVE_INST_VALU', 0.0) / wave
        d['wave_cycles_lds_issue_stall'] = s.get('SQ_WAIT_INST_LDS', 0.0) / wave
    if gui:
        d['mfma_pipe_busy'] = s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 256 * 4)      # of 1024 SIMDs
        d['avg_dispatch_us_at_2p4GHz'] = gui / n / 2400.0
    out[fam] = d
print(json.dumps(out, indent=1))
