"""Two-group recurrence at the decoder's full size: run-to-run bit identity (a race would show here) and agreement with the
one-group kernel on the same input halves (row-major entry)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.layers import FourierEmbedding
dev = torch.device('cuda:0')
torch.manual_seed(0)
mode = ops.MATMUL_SPLIT_F16
emb = FourierEmbedding(2, 128, 64).to(dev).eval()
gru = torch.nn.GRU(128, 128).to(dev).eval()
h0 = torch.randn(128, device=dev) * 0.3
with torch.no_grad():
    ws, bs = ops.fold_linear(list(gru.weight_ih_l0.view(3, 128, 128)), list(gru.bias_ih_l0.view(3, 128)), emb.to_out[2].weight, emb.to_out[2].bias)
    wt = ops.pack_chain_matrices(ws, mode=mode)
    whh = ops.pack_gru_whh(gru.weight_hh_l0, h0, mode=mode)
    bhh = gru.bias_hh_l0.contiguous()
    for n_seq in (96000, 9601):
        x = torch.randn(n_seq * 60, 2, device=dev) * 20
        outs = [ops.trajectory_gru(x, 2, 60, emb.packed_weights(), wt, bs, True, whh, bhh, h0, mode=mode) for _ in range(4)]
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        gi = ops.fourier_linear(x, 2, emb.packed_weights(), wt, bs, folded=True, mode=mode)
        ref = ops.gru_recurrence(gi[0], gi[1], gi[2], 60, whh, bhh, h0, mode=mode)
        print('n_seq %6d: repeat runs bit-identical %s, max |two-group - one-group| %.2e, finite %s' % (
            n_seq, same, float((outs[0] - ref).abs().max()), bool(torch.isfinite(outs[0]).all())))
        assert same and float((outs[0] - ref).abs().max()) < 5e-6
