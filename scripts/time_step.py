"""Where does a cfg3 step go on the host?  (GPU box only)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _bootstrap import ensure_package
ensure_package()
import bench
from tempenslearnagg_qcnet_amd import sharding
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch

dev = torch.device('cuda:0')
cfg = config_scene('cfg3')
model = bench.build_model(cfg['hparams'], dev)
ids = list(range(cfg['batch']))
data = make_batch(ids, **cfg['scene']).to(dev)
with torch.no_grad():
    for _ in range(2):
        pred = model.predict(data)
    torch.cuda.synchronize()
    for it in range(3):
        t0 = time.perf_counter()
        pred = model.predict(data)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rec = sharding.focal_records(ids, data, pred)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print('predict enqueue %.1f ms, drain %.1f ms, focal_records %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    if len(sys.argv) > 1:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            pred = model.predict(data)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=40))
