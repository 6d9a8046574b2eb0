"""Microbenchmark of the FourierEmbedding kernel alone (GPU box only): python scripts/dbg/fourier_bench.py [rows]
QCN_LIB=path loads a timing variant built by scripts/dbg/build_variant.sh (results may be wrong, times only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd import ops, _lib
if os.environ.get('QCN_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['QCN_LIB'])
from tempenslearnagg_qcnet_amd.layers import FourierEmbedding

dev = torch.device('cuda:0')
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800000
four = {d: FourierEmbedding(d, 128, 64).to(dev).eval() for d in (2, 3, 4)}
with torch.no_grad():
    for d in (2, 3, 4):
        xi = torch.randn(n, d, device=dev)
        for norm in (False, True, 'split'):
            f = lambda: four[d](xi, normalized_output=norm)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print('%-8s fourier<%d> norm %-5s rows %8d  %9.1f us  %6.1f TF/s' % (os.environ.get('QCN_TAG', ''), d, norm, n, ms * 1e3,
                                                                         n * (d * 256 * 257 + 32768) / ms * 1e-9))
