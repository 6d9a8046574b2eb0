"""Trajectory-embedding chain (y_emb + GRU input half): qcn_fourier_linear_f32 vs qcn_fourier_folded_linear_f32 (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd import ops, _lib
if os.environ.get("QCN_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["QCN_LIB"])
from tempenslearnagg_qcnet_amd.layers import FourierEmbedding
dev = torch.device('cuda:0')
m = FourierEmbedding(2, 128, 64).to(dev).eval()
ws = [torch.randn(128, 128, device=dev) * 0.1 for _ in range(3)]
bs = [torch.randn(128, device=dev) * 0.1 for _ in range(3)]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    lin = m.to_out[2]
    w_f, b_f = ops.fold_linear(ws, bs, lin.weight, lin.bias)
    wt, bt = ops.pack_chain_matrices(ws), torch.stack(bs).contiguous()
    wtf = ops.pack_chain_matrices(w_f)
    for n in (576000, 5760000):
        x = torch.randn(n, 2, device=dev) * 20
        for rep in range(2):
            print('rows %8d plain %8.1f us  folded %8.1f us' % (n, timeit(lambda: ops.fourier_linear(x, 2, m.packed_weights(), wt, bt)),
                                                            timeit(lambda: ops.fourier_linear(x, 2, m.packed_weights(), wtf, b_f, folded=True))))
