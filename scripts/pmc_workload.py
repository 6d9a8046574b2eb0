"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one pass each): a calibration launch with a known
byte count in the attention kernel's own access width (8 B per lane: qcn_row_normalize_f32 reads and writes N x 512 B),
then one warm cfg3 step.  scripts/pmc_summary.py turns the two counter CSVs into per-launch HBM traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _bootstrap import ensure_package
ensure_package()
import bench
from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch

dev = torch.device('cuda:0')
cfg = config_scene('cfg3')
model = bench.build_model(cfg['hparams'], dev)
data = make_batch(list(range(cfg['batch'])), **cfg['scene']).to(dev)
with torch.no_grad():
    model.predict(data)
    torch.cuda.synchronize()
    x = torch.randn(4 << 20, 128, device=dev)          # 2 GiB: far past the 256 MiB Infinity Cache
    for _ in range(3):
        ops.row_normalize(x)
    torch.cuda.synchronize()
    del x
    model.predict(data)
    torch.cuda.synchronize()
print('done')
