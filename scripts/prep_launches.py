"""Which preparation section launches what: kernel / copy launches per run_prep section of one warm cfg3 step (GPU box only)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _bootstrap import ensure_package
ensure_package()
import bench
from tempenslearnagg_qcnet_amd.utils import prep
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch
from torch.profiler import profile, ProfilerActivity, record_function

dev = torch.device('cuda:0')
cfg = config_scene('cfg3')
model = bench.build_model(cfg['hparams'], dev)
data = make_batch(list(range(cfg['batch'])), **cfg['scene']).to(dev)
orig = prep.run_prep
counter = [0]


def labelled(device, fn):
    import inspect
    frame = inspect.stack()[1]
    label = 'PREP:%s:%d#%d' % (os.path.basename(frame.filename), frame.lineno, counter[0])
    counter[0] += 1
    with record_function(label):
        return orig(device, fn)


with torch.no_grad():
    for _ in range(2):
        model.predict(data)
    torch.cuda.synchronize()
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, 'run_prep', None) is orig:
            mod.run_prep = labelled
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        with record_function('PREP:whole_predict#-1'):
            model.predict(data)
        torch.cuda.synchronize()
ev = prof.events()
ranges = [(e.name, e.time_range.start, e.time_range.end) for e in ev if e.name.startswith('PREP:')]
launch = [e for e in ev if 'Launch' in e.name or 'Memcpy' in e.name or 'Memset' in e.name]
ops_ = [e for e in ev if e.name.startswith('aten::') or e.name.startswith('qcnet_hip::')]
seen = set()
for name, a, b in sorted(ranges, key=lambda r: r[1]):
    if name in seen:
        continue
    seen.add(name)
    n = sum(1 for e in launch if a <= e.time_range.start <= b)
    c = collections.Counter(e.name for e in ops_ if a <= e.time_range.start <= b and e.cpu_parent is not None and e.cpu_parent.name.startswith('PREP:'))
    print('%-50s %4d launches, %6.2f ms host; top-level ops: %s' % (name, n, (b - a) / 1e3, ', '.join('%s x%d' % kv for kv in c.most_common(40))))
    # GPU time of the section's top-level ops (their kernels, whichever stream they ran on)
    g = collections.Counter()
    for e in ops_:
        if a <= e.time_range.start <= b and e.cpu_parent is not None and e.cpu_parent.name.startswith('PREP:'):
            g[e.name] += getattr(e, 'device_time_total', None) or getattr(e, 'cuda_time_total', 0)
    print('%-50s GPU us by op: %s' % ('', ', '.join('%s %.0f' % kv for kv in g.most_common(14))))
