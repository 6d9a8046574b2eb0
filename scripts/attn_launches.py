"""Per-launch roofline of the positional attention launches of one warm step (isolated: preparation stream off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap
_bootstrap.ensure_package()
from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.utils import prep
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch
import bench

cfg = config_scene(sys.argv[1] if len(sys.argv) > 1 else 'cfg3')
dev = torch.device('cuda:0')
model = bench.build_model(cfg['hparams'], dev)
data = make_batch(list(range(cfg['batch'])), **cfg['scene']).to(dev)
prep.ENABLED = False
model.overlap_aggregate = False
with torch.no_grad():
    for _ in range(2):
        model.predict(data)
    torch.cuda.synchronize()
    ops.prof_enable(True, kinds=[ops.PROF_ATTN_FUSED])
    model.predict(data)
    torch.cuda.synchronize()
    ops.prof_enable(False)
tot_b = tot_ms = 0.0
for kind, units, e0, e1 in ops._PROF_LOG:
    ms = e0.elapsed_time(e1)
    tot_b += units; tot_ms += ms
    print('%8.1f us  %7.1f MB  %5.2f TB/s  frac %.3f' % (ms * 1e3, units / 1e6, units / ms / 1e9, units / ms / 1e9 / 8.0))
print('total %.2f ms, %.2f TB/s, frac %.3f' % (tot_ms, tot_b / tot_ms / 1e9, tot_b / tot_ms / 1e9 / 8.0))
