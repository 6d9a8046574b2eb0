"""Load imbalance of qcn_attn_modes_f32's static row partition (256 equal row chunks) on the cfg3 graphs: tiles (16 edges)
per chunk, max / mean, against an edge-balanced partition.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
orig = ops.attn_modes
seen = {}


def spy(q, k, v, rs, rowptr, col, eid, wpack, bv, rep, layout, col_shift=0):
    key = (rowptr.numel() - 1, col.numel(), rep)
    if key not in seen:
        rp = rowptr.cpu().long()
        tiles = (rp[1:] - rp[:-1] + 15) // 16
        n = tiles.numel()
        nb = min(n, 256)
        per = (n + nb - 1) // nb
        chunks = [int(tiles[i * per:(i + 1) * per].sum()) for i in range(nb)]
        total = int(tiles.sum())
        # edge-balanced: chunk b = rows whose first tile index lies in [b T / nb, (b + 1) T / nb)
        cum = torch.cat([torch.zeros(1, dtype=torch.long), tiles.cumsum(0)])
        bounds = [int(torch.searchsorted(cum, torch.tensor(b * total // nb))) for b in range(nb)] + [n]
        bal = [int(cum[bounds[b + 1]] - cum[bounds[b]]) for b in range(nb)]
        seen[key] = (max(chunks) / (total / nb), max(bal) / (total / nb), total)
    return orig(q, k, v, rs, rowptr, col, eid, wpack, bv, rep, layout, col_shift)


ops.attn_modes = spy
import tempenslearnagg_qcnet_amd.layers.attention_layer as al
if hasattr(al, 'ops'):
    al.ops.attn_modes = spy
with torch.no_grad():
    model.predict(data)
torch.cuda.synchronize()
for key, (a, b, t) in seen.items():
    print('rows %6d edges %8d rep %d tiles %7d: equal-row chunks max/mean %.3f, tile-balanced chunks %.3f' % (key + (t, a, b)))
