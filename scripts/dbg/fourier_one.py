"""One FourierEmbedding launch (GPU box only): python scripts/dbg/fourier_one.py [input_dim] [rows] [norm]"""
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
d = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 800000
norm = {'0': False, '1': True, '2': 'split'}[sys.argv[3] if len(sys.argv) > 3 else '1']
m = FourierEmbedding(d, 128, 64).to(dev).eval()
with torch.no_grad():
    x = torch.randn(n, d, device=dev)
    for _ in range(3):
        m(x, normalized_output=norm)
    torch.cuda.synchronize()
