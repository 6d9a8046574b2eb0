import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import _bootstrap; _bootstrap.ensure_package()
import torch
from helpers import c_attention, rel_err
from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.utils.graph import CSRGraph
from test_gpu_attention import _graph, _feat, _pos_weights, _materialise, _fused
DEV='cuda:0'
n_src, n_dst, E = 40, 30, 500
ei = _graph(n_src, n_dst, E, 31)
q, k, v, r = _feat(n_dst, 1, 1.5), _feat(n_src, 2, 1.5), _feat(n_src, 3), _feat(E, 4, 2.0) + 0.3
w_kr, w_vr, b_vr, gamma, beta = _pos_weights(9)
graph = CSRGraph.from_edge_index(ei.to(DEV), n_src, n_dst)
def run(name, w_kr, w_vr, b_vr, gamma, beta):
    kr, vr = _materialise(r, w_kr, w_vr, b_vr, gamma, beta)
    want = c_attention(q, k, v, kr, vr, ei)
    got = _fused(q, k, v, r, graph, w_kr, w_vr, b_vr, gamma, beta)
    print('%-28s rel_err %.3e' % (name, rel_err(got, want)))
    return got.cpu(), want
z = torch.zeros_like(w_kr); zb = torch.zeros_like(b_vr); one = torch.ones_like(gamma)
run('all zero', z, z, zb, one, zb)
run('bias only', z, z, b_vr, one, zb)
run('wv only', z, w_vr, zb, one, zb)
run('wk only', w_kr, z, zb, one, zb)
run('wk only head0', torch.cat([w_kr[:16], z[16:]]), z, zb, one, zb)
got, want = run('wk one row', torch.cat([w_kr[:1], z[1:]]), z, zb, one, zb)
run('full', w_kr, w_vr, b_vr, gamma, beta)
g, w = run('wv only', z, w_vr, zb, one, zb)
print((g-w).abs().max(dim=1).values[:10])
print(g[0,:8], w[0,:8])
