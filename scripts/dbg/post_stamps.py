"""attn_post phase stamps (diagnostic build: scripts/dbg/build_variant.sh stamps -DQCN_CHAIN_STAMPS; GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd import ops, _lib
_lib.LIB_PATH = os.path.abspath(os.environ['QCN_LIB'])
from tempenslearnagg_qcnet_amd.layers import AttentionLayer
dev = torch.device('cuda:0')
layer = AttentionLayer(128, 8, 16, 0.0, bipartite=False, has_pos_emb=True).to(dev).eval()
with torch.no_grad():
    packs = layer._kernel_packs()
    for n in (256, 65536, 800000):
        x = [torch.randn(n, 128, device=dev) for _ in range(4)]
        for _ in range(3):
            ops.attn_post(x[0], x[1], x[2], x[3], packs['post'])
        torch.cuda.synchronize()
