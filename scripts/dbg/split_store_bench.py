"""FourierEmbedding output formats: fp32 rows, row-normalised fp32 rows, split-f16 rows (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd.layers import FourierEmbedding
dev = torch.device('cuda:0')
m = FourierEmbedding(3, 128, 64).to(dev).eval()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    for n in (480000, 2180000):
        x = torch.randn(n, 3, device=dev)
        for mode in (False, True, 'split'):
            print('rows %8d normalized_output=%-6s %8.1f us' % (n, mode, timeit(lambda: m(continuous_inputs=x, categorical_embs=None, normalized_output=mode))))
