"""Microbenchmark of the MFMA chain kernels (attn_post / ln_linear / fourier) at the row counts of cfg3.
Usage: python scripts/bench_nodeops.py [rows ...]   (GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd import ops, _lib
if os.environ.get('QCN_LIB'):      # a timing variant built by scripts/dbg/build_variant.sh (results are wrong, times only)
    _lib.LIB_PATH = os.path.abspath(os.environ['QCN_LIB'])
from tempenslearnagg_qcnet_amd.layers import AttentionLayer, FourierEmbedding

dev = torch.device('cuda:0')
torch.manual_seed(0)
layer = AttentionLayer(128, 8, 16, 0.0, bipartite=False, has_pos_emb=True).to(dev).eval()
four = {d: FourierEmbedding(d, 128, 64).to(dev).eval() for d in (2, 3, 4)}
rows_list = [int(a) for a in sys.argv[1:]] or [9600, 96000, 800000]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    packs = layer._kernel_packs()
    for n in rows_list:
        x = [torch.randn(n, 128, device=dev) for _ in range(4)]
        ms = timeit(lambda: ops.attn_post(x[0], x[1], x[2], x[3], packs['post']))
        print('attn_post  rows %8d  %9.1f us  %6.1f TF/s' % (n, ms * 1e3, n * 11 * 32768 / ms * 1e-9))
        ln = layer.attn_prenorm_x_src
        for n_next, key in ((0, None), (1, 'pre_q'), (3, 'pre_qkv')):
            nx = (None, None, None, None) if key is None else packs[key] + (ln.weight, ln.bias)
            ms = timeit(lambda: ops.attn_post_chain(x[0], x[3], packs['post_chain'], *nx))
            print('attn_post_chain+%d rows %8d  %9.1f us  %6.1f TF/s' % (n_next, n, ms * 1e3, n * (12 + n_next) * 32768 / ms * 1e-9))
        ms = timeit(lambda: ops.ln_linear(x[0], ln.weight, ln.bias, *packs['pre_all'], want_xn=True))
        print('ln_linear4 rows %8d  %9.1f us  %6.1f TF/s' % (n, ms * 1e3, n * 4 * 32768 / ms * 1e-9))
        for d in (2, 3, 4):
            xi = torch.randn(n, d, device=dev)
            ms = timeit(lambda: four[d](xi))
            print('fourier<%d> rows %8d  %9.1f us  %6.1f TF/s' % (d, n, ms * 1e3, n * (d * 256 * 257 + 32768) / ms * 1e-9))
