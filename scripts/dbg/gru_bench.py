"""qcn_gru_blocked_f32 / ops.trajectory_gru at the decoder's sizes (96 000 and 9 600 sequences x 60 steps): us per launch pair."""
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
emb = FourierEmbedding(2, 128, 64).to(dev).eval()
gru = torch.nn.GRU(128, 128).to(dev).eval()
h0 = torch.randn(128, device=dev) * 0.3
mode = ops.MATMUL_SPLIT_F16
with torch.no_grad():
    ws, bs = ops.fold_linear(list(gru.weight_ih_l0.view(3, 128, 128)), list(gru.bias_ih_l0.view(3, 128)), emb.to_out[2].weight, emb.to_out[2].bias)
    wt = ops.pack_chain_matrices(ws, mode=mode)
    bs = (bs if torch.is_tensor(bs) else torch.stack(bs)).contiguous()
    whh = ops.pack_gru_whh(gru.weight_hh_l0, h0, mode=mode)
    bhh = gru.bias_hh_l0.contiguous()
    for n_seq in (9600, 96000):
        x = torch.randn(n_seq * 60, 2, device=dev) * 20
        rows = int(_lib.lib().qcn_gru_blocked_rows(n_seq, 60))
        gi = torch.randn(3, rows, 128, device=dev)
        h = torch.empty(n_seq, 128, device=dev)
        def rec():
            _lib.check(_lib.lib().qcn_gru_blocked_f32(ops._ptr(gi[0]), ops._ptr(gi[1]), ops._ptr(gi[2]), ops._ptr(whh), ops._ptr(bhh),
                                                      ops._ptr(h0), n_seq, 60, ops._ptr(h), ops._stream(dev)), 'gru')
        for name, fn in (('recurrence', rec), ('chain + recurrence', lambda: ops.trajectory_gru(x, 2, 60, emb.packed_weights(), wt, bs, True, whh, bhh, h0, mode=mode))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            print('%-20s n_seq %6d  %8.1f us' % (name, n_seq, e0.elapsed_time(e1) / 5 * 1e3))
