"""SQ counters of one rocprofv3 --pmc pass, per kernel name prefix: sq_one.py <dir> [substring]"""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*counter_collection.csv'))[0]
sub = sys.argv[2] if len(sys.argv) > 2 else 'fourier'
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if sub in k:
        name = k.split('(')[0][-60:]
        agg[name][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            cnt[name] += 1
for name, s in agg.items():
    wave = s['SQ_WAVE_CYCLES']
    gui = s['GRBM_GUI_ACTIVE'] / 8.0
    print(name, 'dispatches', cnt[name])
    print('   parked %.3f  issue-stalled %.3f  issuing %.3f (valu %.3f)  lds-stall %.3f  mfma-busy %.3f  us/dispatch@2.4GHz %.1f  insts_valu/wave-quadcycle %.3f'
          % (s['SQ_WAIT_ANY'] / wave, s['SQ_WAIT_INST_ANY'] / wave, s['SQ_ACTIVE_INST_ANY'] / wave, s['SQ_ACTIVE_INST_VALU'] / wave,
             s['SQ_WAIT_INST_LDS'] / wave, s['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024) if gui else 0, gui / max(cnt[name], 1) / 2400.0,
             s.get('SQ_INSTS_VALU', 0) / wave))
