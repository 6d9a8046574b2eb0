"""Capture one full predict (ten frames + aggregate) of a resident cfg3 batch into a HIP graph and replay it: the graph
construction of the batch (host round trips) is done once, eagerly, and handed out again by QCN_PREP_MEMO=1."""
import os, sys, time
os.environ['QCN_PREP_MEMO'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _bootstrap import ensure_package
ensure_package()
import bench
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch

dev = torch.device('cuda:0')
cfg = config_scene(sys.argv[1] if len(sys.argv) > 1 else 'cfg3')
model = bench.build_model(cfg['hparams'], dev)
data = make_batch(list(range(cfg['batch'])), **cfg['scene']).to(dev)
keys = ('traj_refine', 'pi_softmax', 'final_m')
s = torch.cuda.Stream(device=dev)
with torch.no_grad(), torch.cuda.stream(s):
    for _ in range(3):
        want = model.predict(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.predict(data)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        got = model.predict(data)
    torch.cuda.synchronize()
    for k in keys:
        got[k].zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in keys:
        print(k, 'bit-identical' if torch.equal(got[k], want[k]) else 'DIFFERS %g' % float((got[k] - want[k]).abs().max()))
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print('eager (memoized sections) %.2f ms/step, replay %.2f ms/step' % (eager * 1e3, (time.perf_counter() - t0) / 10 * 1e3))
