"""Shapes and per-launch times of every qcn_attn_fused_f32 call of one cfg3 step (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _bootstrap import ensure_package
ensure_package()
import bench
from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch
import tempenslearnagg_qcnet_amd.layers.attention_layer as al

dev = torch.device('cuda:0')
cfg = config_scene('cfg3')
model = bench.build_model(cfg['hparams'], dev)
data = make_batch(list(range(cfg['batch'])), **cfg['scene']).to(dev)
log = []
orig = ops.attn_fused


def wrapped(q, k, v, rhat, rowptr, col, eid, wk, wv, bv, rep, layout, col_shift, row_index=None, src_offset=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(q, k, v, rhat, rowptr, col, eid, wk, wv, bv, rep, layout, col_shift, row_index, src_offset)
    e1.record()
    log.append((q.size(0), k.size(0), rhat.size(0), col.numel(), rep, layout, eid is not None, row_index is not None, e0, e1))
    return out


with torch.no_grad():
    model.predict(data)
    torch.cuda.synchronize()
    al.ops.attn_fused = wrapped
    model.predict(data)
    torch.cuda.synchronize()
tot = 0.0
print('  n_dst    n_src  rhat_rows  col_len rep layout eid group      us   edge-visits  GB/s(512B*visits)  ns/visit/CU')
for n_dst, n_src, nr, ncol, rep, layout, eid, grp, e0, e1 in log:
    ms = e0.elapsed_time(e1)
    tot += ms
    visits = ncol * rep if not grp else None
    print('%8d %8d %9d %9d %3d %3d %5s %5s %8.1f %12s' % (n_dst, n_src, nr, ncol, rep, layout, eid, grp, ms * 1e3, visits))
print('total ms', tot)
