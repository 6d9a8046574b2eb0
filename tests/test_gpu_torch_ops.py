"""GPU: every kernel entry point through the dispatcher — ``torch.ops.qcnet_hip.<name>``.

One test per registered op: called through ``torch.ops`` with positional schema arguments and checked against a plain
torch fp32 / fp64 form of the same arithmetic (or, for the integer ops, bit-exactly).  The last tests check that the set of
registered ops is the set tested here and that the ops trace under FakeTensor (the fake implementations)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import c_attention, c_radius, rel_err

from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.layers import FourierEmbedding

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
Q = torch.ops.qcnet_hip
TESTED = set()


def op(name):
    TESTED.add(name)
    return getattr(Q, name)


def _graph(n_src, n_dst, E, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst, (E,), generator=g)])


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_op_radius_pairs():
    x, y = _rand(60, 2, seed=1, scale=40.0), _rand(90, 2, seed=2, scale=40.0)
    px, py = torch.tensor([0, 25, 60]), torch.tensor([0, 50, 90])
    ei, off = op('radius_pairs')(x.to(DEV), y.to(DEV), 30.0, px.to(DEV), py.to(DEV), 8, False, None, None, True)
    assert torch.equal(ei.cpu(), c_radius(x, y, 30.0, px, py, 8))
    assert int(off[-1]) == ei.size(1)


def test_op_csr_build_and_static():
    ei = _graph(40, 30, 700, 3).to(DEV)
    rowptr, col, eid = op('csr_build')(ei, 30)
    order = torch.argsort(ei[1], stable=True)
    assert torch.equal(eid.long(), order) and torch.equal(col.long(), ei[0][order])
    assert torch.equal(rowptr.long(), torch.cat([ei.new_zeros(1), torch.bincount(ei[1], minlength=30).cumsum(0)]))
    outs = [torch.empty_like(rowptr), torch.empty_like(col), torch.empty_like(eid)]
    assert op('csr_build_static')(ei, 30, *outs) is None
    for a, b in zip(outs, (rowptr, col, eid)):
        assert torch.equal(a, b)


def test_op_attn_fwd():
    ei = _graph(40, 30, 500, 4)
    q, k, v = _rand(30, 128, seed=5), _rand(40, 128, seed=6), _rand(40, 128, seed=7)
    kr, vr = _rand(500, 128, seed=8, scale=0.5), _rand(500, 128, seed=9, scale=0.5)
    rowptr, col, eid = Q.csr_build(ei.to(DEV), 30)
    got = op('attn_fwd')(q.to(DEV), k.to(DEV), v.to(DEV), kr.to(DEV), vr.to(DEV), rowptr, col, eid)
    assert rel_err(got, c_attention(q, k, v, kr, vr, ei)) < 1e-5


def test_op_row_normalize_and_split_rows():
    x = _rand(333, 128, seed=10, scale=3.0).to(DEV)
    y = op('row_normalize')(x, 1e-5)
    assert rel_err(y, F.layer_norm(x, (128,), None, None, 1e-5)) < 1e-6
    sp = op('split_rows')(y, 11)
    assert sp.dtype == torch.float16 and tuple(sp.shape) == (333, 256)
    back = (sp[:, :128].double() + sp[:, 128:].double()) / 2048.0
    assert float((back - y.double()).abs().max()) <= 2.0 ** -21 * float(y.abs().max())


@pytest.mark.parametrize('mode', [ops.MATMUL_EXACT_F32, ops.MATMUL_SPLIT_F16])
def test_op_ln_linear(mode):
    x = _rand(500, 128, seed=11, scale=2.0).to(DEV)
    g, b = (1.0 + _rand(128, seed=12, scale=0.1)).to(DEV), _rand(128, seed=13, scale=0.1).to(DEV)
    ws = [_rand(128, 128, seed=14 + j, scale=0.1).to(DEV) for j in range(3)]
    bias = _rand(3, 128, seed=20, scale=0.1).to(DEV)
    outs = op('ln_linear')(x, g, b, ops.pack_chain_matrices(ws, mode=mode), bias, True, mode)
    assert len(outs) == 4
    xn = F.layer_norm(x.double(), (128,), g.double(), b.double(), 1e-5)
    assert rel_err(outs[3], xn) < 1e-6
    for j in range(3):
        assert rel_err(outs[j], xn @ ws[j].double().t() + bias[j].double()) < 1e-5


def _torch_attn_post(agg, s, x, p):
    xn = F.layer_norm(x, (128,), p['pre_g'], p['pre_b'], 1e-5)
    g = torch.sigmoid(F.linear(torch.cat([agg, xn], -1), p['wg'], p['bg']))
    u = agg + g * (s - agg)
    x1 = x + F.layer_norm(F.linear(u, p['wo'], p['bo']), (128,), p['lp_g'], p['lp_b'], 1e-5)
    h = F.relu(F.linear(F.layer_norm(x1, (128,), p['lf_g'], p['lf_b'], 1e-5), p['w1'], p['b1']))
    return x1 + F.layer_norm(F.linear(h, p['w2'], p['b2']), (128,), p['lq_g'], p['lq_b'], 1e-5)


def test_op_attn_post():
    mode = ops.MATMUL_SPLIT_F16
    mk = lambda *s, seed, scale=0.1: _rand(*s, seed=seed, scale=scale).double()
    p = dict(wg=mk(128, 256, seed=30), bg=mk(128, seed=31), wo=mk(128, 128, seed=32), bo=mk(128, seed=33),
             w1=mk(512, 128, seed=34), b1=mk(512, seed=35), w2=mk(128, 512, seed=36), b2=mk(128, seed=37))
    for i, n in enumerate(('lp', 'lf', 'lq', 'pre')):
        p[n + '_g'] = 1.0 + mk(128, seed=40 + i)
        p[n + '_b'] = mk(128, seed=50 + i)
    f = lambda t: t.float().to(DEV)
    mats = [f(p['wg'][:, :128]), f(p['wg'][:, 128:]), f(p['wo'])]
    for hb in range(4):
        mats += [f(p['w1'][128 * hb:128 * hb + 128]), f(p['w2'][:, 128 * hb:128 * hb + 128])]
    packed, scales = ops.pack_chain_set(mats, mode=mode)
    vecs = [p['bg'], p['bo'], p['lp_g'], p['lp_b'], p['lf_g'], p['lf_b'], p['b1'], p['b2'], p['lq_g'], p['lq_b'], p['pre_g'],
            p['pre_b']]
    pack = torch.cat([packed.reshape(-1)] + [f(v).reshape(-1) for v in vecs] + [ops.scale_table(scales)]).contiguous()
    agg, s, x = (_rand(700, 128, seed=60 + i).double() for i in range(3))
    got = op('attn_post')(f(agg), None, f(s), f(x), pack, mode)
    assert rel_err(got, _torch_attn_post(agg, s, x, p)) < 1e-5


@pytest.mark.parametrize('mode,n_next,n_rows', [(ops.MATMUL_SPLIT_F16, 0, 700), (ops.MATMUL_SPLIT_F16, 1, 700),
                                               (ops.MATMUL_SPLIT_F16, 3, 77), (ops.MATMUL_EXACT_F32, 3, 300),
                                               (ops.MATMUL_BF16, 1, 130)])
def test_op_attn_post_chain(mode, n_next, n_rows):
    """qcn_attn_post_chain_f32: to_s inside the kernel and the next layer's LayerNorm + projections on the output row, against
    the fp64 torch form of layers/attention_layer.py:83-84,106-108,119-122 (+ :74-79 of the following call)."""
    mk = lambda *s, seed, scale=0.1: _rand(*s, seed=seed, scale=scale).double()
    p = dict(wg=mk(128, 256, seed=30), bg=mk(128, seed=31), wo=mk(128, 128, seed=32), bo=mk(128, seed=33),
             w1=mk(512, 128, seed=34), b1=mk(512, seed=35), w2=mk(128, 512, seed=36), b2=mk(128, seed=37),
             ws=mk(128, 128, seed=38), bs=mk(128, seed=39))
    for i, n in enumerate(('lp', 'lf', 'lq', 'pre', 'nx')):
        p[n + '_g'] = 1.0 + mk(128, seed=40 + i)
        p[n + '_b'] = mk(128, seed=50 + i)
    f = lambda t: t.float().to(DEV)
    pack = ops.pack_attn_post_chain(f(p['wg']), f(p['bg']), f(p['ws']), f(p['bs']), f(p['wo']), f(p['bo']), f(p['w1']), f(p['b1']),
                                    f(p['w2']), f(p['b2']), (f(p['lp_g']), f(p['lp_b'])), (f(p['lf_g']), f(p['lf_b'])),
                                    (f(p['lq_g']), f(p['lq_b'])), (f(p['pre_g']), f(p['pre_b'])), mode=mode)
    agg, x = (_rand(n_rows, 128, seed=60 + i).double() for i in range(2))
    nw = [mk(128, 128, seed=80 + j) for j in range(n_next)]
    nb = mk(max(n_next, 1), 128, seed=90)[:n_next]
    args = [None] * 4
    if n_next:
        args = [ops.pack_chain_matrices([f(w) for w in nw], mode=mode), f(nb), f(p['nx_g']), f(p['nx_b'])]
    nsc = ops.static_operand_scale(ops.layer_norm_bound(p['nx_g'], p['nx_b'])) if n_next == 3 else 0.0   # both operand forms
    outs = op('attn_post_chain')(f(agg), f(x), pack, *args, nsc, mode)
    assert len(outs) == 1 + n_next
    if mode == ops.MATMUL_BF16:   # every matrix operand rounded to bf16: compare loosely with the fp64 form (exercises the path)
        tol = 3e-2
    else:
        tol = 1e-5
    s = F.linear(F.layer_norm(x, (128,), p['pre_g'], p['pre_b'], 1e-5), p['ws'], p['bs'])
    want = _torch_attn_post(agg, s, x, p)
    assert rel_err(outs[0], want) < tol
    wn = F.layer_norm(outs[0].double().cpu() if mode == ops.MATMUL_BF16 else want, (128,), p['nx_g'], p['nx_b'], 1e-5)
    for j in range(n_next):
        assert rel_err(outs[1 + j], wn @ nw[j].t() + nb[j]) < (2e-2 if mode == ops.MATMUL_BF16 else 2e-5)


def test_op_proposal_cumsum():
    """qcn_proposal_cumsum_f32 against the torch forms of modules/qcnet_decoder.py:234-250."""
    import math
    chunks = [_rand(333, 40, seed=200 + i).to(DEV) for i in range(3)]
    x = torch.cat([c.cpu() for c in chunks], dim=-1).view(333, 60, 2)
    got = op('proposal_cumsum')(chunks, 2, ops.CUMSUM_IDENTITY, 0.0, False)
    assert tuple(got.shape) == (333, 60, 2) and rel_err(got, torch.cumsum(x, dim=-2)) < 2e-6
    got = op('proposal_cumsum')(chunks, 2, ops.CUMSUM_ELU1, 0.1, False)
    assert rel_err(got, torch.cumsum(F.elu(x) + 1.0, dim=-2) + 0.1) < 2e-6
    heads = [_rand(77, 20, seed=210 + i).to(DEV) for i in range(3)]
    xh = torch.cat([c.cpu() for c in heads], dim=-1).unsqueeze(-1)
    got = op('proposal_cumsum')(heads, 1, ops.CUMSUM_TANH_PI, 0.0, False)
    assert tuple(got.shape) == (77, 60, 1) and rel_err(got, torch.cumsum(torch.tanh(xh) * math.pi, dim=-2)) < 2e-6
    got = op('proposal_cumsum')(heads, 1, ops.CUMSUM_ELU1, 0.02, True)
    assert rel_err(got, 1.0 / (torch.cumsum(F.elu(xh) + 1.0, dim=-2) + 0.02)) < 2e-6


def test_op_fourier_embed_and_fourier_linear():
    m = FourierEmbedding(input_dim=2, hidden_dim=128, num_freq_bands=64).eval().to(DEV)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if prm.dim() == 1:
                prm.add_(torch.randn_like(prm) * 0.1)
    mode = ops.matmul_mode()
    x = _rand(900, 2, seed=70, scale=10.0).to(DEV)
    with torch.no_grad():
        y = op('fourier_embed')(x, 2, m.packed_weights(), None, 1, 0, 900, mode)
        # plain torch form of layers/fourier_embedding.py:49-67 in fp64
        xd = x.double()
        ang = xd.unsqueeze(-1) * m.freqs.weight.double() * 2 * torch.pi
        feat = torch.cat([ang.cos(), ang.sin(), xd.unsqueeze(-1)], -1)
        acc = 0
        for i in range(2):
            lin1, ln, _, lin2 = m.mlps[i]
            h = F.relu(F.layer_norm(F.linear(feat[:, i], lin1.weight.double(), lin1.bias.double()), (128,),
                                    ln.weight.double(), ln.bias.double(), 1e-5))
            acc = acc + F.linear(h, lin2.weight.double(), lin2.bias.double())
        ln, _, lin = m.to_out
        want = F.linear(F.relu(F.layer_norm(acc, (128,), ln.weight.double(), ln.bias.double(), 1e-5)), lin.weight.double(),
                        lin.bias.double())
        assert rel_err(y, want) < 1e-5
        ws = [_rand(128, 128, seed=71 + j, scale=0.1).to(DEV) for j in range(2)]
        bias = _rand(2, 128, seed=75, scale=0.1).to(DEV)
        ys = op('fourier_linear')(x, 2, m.packed_weights(), ops.pack_chain_matrices(ws, mode=mode), bias, False, mode)
    for j in range(2):
        assert rel_err(ys[j], want @ ws[j].double().t() + bias[j].double()) < 1e-5


def test_op_rel_geometry():
    g = torch.Generator().manual_seed(80)
    ps, pd = torch.randn(50, 3, generator=g) * 30, torch.randn(40, 2, generator=g) * 30
    hs, hd = (torch.rand(50, generator=g) * 2 - 1) * 3.14, (torch.rand(40, generator=g) * 2 - 1) * 3.14
    ei = _graph(50, 40, 600, 81)
    out = op('rel_geometry')(ps.to(DEV), pd.to(DEV), hs.to(DEV), hd.to(DEV), ei.to(DEV), None).cpu()
    dp = ps[ei[0], :2] - pd[ei[1]]
    hv = torch.stack([hd[ei[1]].cos(), hd[ei[1]].sin()], -1)
    ang = torch.atan2(hv[:, 0] * dp[:, 1] - hv[:, 1] * dp[:, 0], (hv * dp).sum(-1))
    dh = hs[ei[0]] - hd[ei[1]]
    wrap = -torch.pi + (dh + torch.pi) % (2 * torch.pi)
    assert torch.allclose(out[:, 0], dp.norm(dim=-1), rtol=1e-6, atol=1e-5)
    assert float(((out[:, 1] - ang + torch.pi) % (2 * torch.pi) - torch.pi).abs().max()) < 1e-5
    assert float(((out[:, 2] - wrap + torch.pi) % (2 * torch.pi) - torch.pi).abs().max()) < 1e-5


def test_op_mlp_heads():
    mode = ops.matmul_mode()
    heads, outs = [], (40, 1)
    for i, od in enumerate(outs):
        heads.append(tuple(t.to(DEV) for t in (_rand(128, 128, seed=90 + i, scale=0.1), _rand(128, seed=92 + i, scale=0.1),
                                               1.0 + _rand(128, seed=94 + i, scale=0.1), _rand(128, seed=96 + i, scale=0.1),
                                               _rand(od, 128, seed=98 + i, scale=0.1), _rand(od, seed=100 + i, scale=0.1))))
    x = _rand(450, 128, seed=102).to(DEV)
    ys = op('mlp_heads')(x, ops.pack_mlp_heads(heads, mode=mode), list(outs), mode)
    for (w1, b1, g, b, w2, b2), y in zip(heads, ys):
        h = F.relu(F.layer_norm(F.linear(x.double(), w1.double(), b1.double()), (128,), g.double(), b.double(), 1e-5))
        assert rel_err(y, F.linear(h, w2.double(), b2.double())) < 1e-5


def test_op_gru_recurrence():
    mode = ops.MATMUL_SPLIT_F16
    torch.manual_seed(110)
    gru = torch.nn.GRU(128, 128).eval()
    with torch.no_grad():
        for prm in gru.parameters():
            prm.copy_(torch.randn_like(prm) * 0.12)
    n_seq, n_steps = 40, 12
    seq, h0 = torch.randn(n_steps, n_seq, 128), torch.randn(128) * 0.3
    with torch.no_grad():
        want = gru(seq, h0.view(1, 1, 128).repeat(1, n_seq, 1))[1].squeeze(0)
        gi = F.linear(seq.transpose(0, 1).reshape(-1, 128), gru.weight_ih_l0, gru.bias_ih_l0).to(DEV)
    gi_r, gi_z, gi_n = (t.contiguous() for t in gi.chunk(3, dim=-1))
    got = op('gru_recurrence')(gi_r, gi_z, gi_n, n_steps, ops.pack_gru_whh(gru.weight_hh_l0.detach().to(DEV), h0.to(DEV), mode=mode),
                               gru.bias_hh_l0.detach().to(DEV), h0.to(DEV), mode)
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize('n_seq,n_steps', [(70, 9), (5, 1), (33, 2), (64, 3), (161, 60)])
def test_op_trajectory_gru(n_seq, n_steps):
    """FourierEmbedding -> GRU input matrices -> GRU recurrence as one op (modules/qcnet_decoder.py:255-257), against
    torch.nn.GRU fed with the fp64 embedding.  The recurrence is the two-group kernel (two blocks of 32 sequences per
    workgroup, half a step apart): one and two steps exercise its prologue / epilogue phases alone, 5 / 33 / 161 sequences the
    workgroup whose second group repeats the last block, 64 the exact fit."""
    mode = ops.MATMUL_SPLIT_F16
    torch.manual_seed(120)
    m = FourierEmbedding(input_dim=2, hidden_dim=128, num_freq_bands=64).eval().to(DEV)
    gru = torch.nn.GRU(128, 128).eval().to(DEV)
    with torch.no_grad():
        for prm in gru.parameters():
            prm.copy_(torch.randn_like(prm) * 0.12)
    x = _rand(n_seq * n_steps, 2, seed=121, scale=5.0).to(DEV)
    h0 = (_rand(128, seed=122) * 0.3).to(DEV)
    with torch.no_grad():
        emb = m(continuous_inputs=x, categorical_embs=None)                       # [n_seq * n_steps, 128], row = seq * n_steps + t
        want = gru.double()(emb.double().view(n_seq, n_steps, 128).transpose(0, 1).contiguous(),
                            h0.double().view(1, 1, 128).repeat(1, n_seq, 1).contiguous())[1].squeeze(0)
        gru.float()
        w_ih = [w.contiguous() for w in gru.weight_ih_l0.chunk(3, dim=0)]
        wt = ops.pack_chain_matrices(w_ih, mode=mode)
        bias = gru.bias_ih_l0.view(3, 128).contiguous()
        whh = ops.pack_gru_whh(gru.weight_hh_l0.detach(), h0, mode=mode)
        got = op('trajectory_gru')(x, 2, n_steps, m.packed_weights(), wt, bias, False, whh, gru.bias_hh_l0.detach(), h0, mode)
    assert got.shape == (n_seq, 128)
    assert rel_err(got, want) < 2e-5


def _pos_case(rep, seed):
    n_rows, n_src, E = 50, 60, 1500
    ei = _graph(n_src, n_rows, E, seed)
    q = _rand(n_rows * rep, 128, seed=seed + 1, scale=1.5)
    k, v = _rand(n_src, 128, seed=seed + 2, scale=1.5), _rand(n_src, 128, seed=seed + 3)
    rhat = F.layer_norm(_rand(E, 128, seed=seed + 4, scale=2.0), (128,), None, None, 1e-5)
    wk, wv = _rand(128, 128, seed=seed + 5, scale=0.1), _rand(128, 128, seed=seed + 6, scale=0.1)
    bv = _rand(128, seed=seed + 7, scale=0.1)
    return ei, q, k, v, rhat, wk, wv, bv


def test_op_attn_fused():
    ei, q, k, v, rhat, wk, wv, bv = _pos_case(1, 120)
    rowptr, col, eid = Q.csr_build(ei.to(DEV), 50)
    got = op('attn_fused')(q.to(DEV), k.to(DEV), v.to(DEV), rhat.to(DEV), rowptr, col, eid, wk.to(DEV), wv.to(DEV), bv.to(DEV),
                           1, ops.LAYOUT_PLAIN, 0, None, None)
    want = c_attention(q, k, v, rhat @ wk.t(), rhat @ wv.t() + bv, ei)
    assert rel_err(got, want) < 2e-5


def test_op_attn_modes():
    K = 6
    ei, q, k, v, rhat, wk, wv, bv = _pos_case(K, 130)
    rowptr, col, eid = Q.csr_build(ei.to(DEV), 50)
    dev = lambda t: t.to(DEV)
    got = op('attn_modes')(dev(q), dev(k), dev(v), ops.split_rows(dev(rhat)), rowptr, col, eid,
                           ops.pack_attn_modes(dev(wk), dev(wv), mode=ops.MATMUL_SPLIT_F16), dev(bv), K, ops.LAYOUT_AGENT_MAJOR, 0)
    kr, vr = rhat @ wk.t(), rhat @ wv.t() + bv
    for mode_id in range(K):                   # agent-major: destination node = row * K + mode
        want = c_attention(q[mode_id::K].contiguous(), k, v, kr, vr, ei)
        assert rel_err(got[mode_id::K], want) < 2e-5, mode_id
    # the storing form returns the same result plus the per-edge projections; the cached form reproduces it from them
    out, kk, vv = op('attn_modes_store')(dev(q), dev(k), dev(v), ops.split_rows(dev(rhat)), rowptr, col, eid,
                                         ops.pack_attn_modes(dev(wk), dev(wv), mode=ops.MATMUL_SPLIT_F16), dev(bv), K,
                                         ops.LAYOUT_AGENT_MAJOR, 0)
    assert torch.equal(out, got) and kk.numel() == vv.numel() == 128 * ops.attn_modes_table_slots(col.numel(), rowptr.numel() - 1)
    kk_rows, vv_rows = ops.attn_modes_table_rows(kk, vv, rowptr)
    assert rel_err(kk_rows, (k[col.long().cpu()] + kr[eid.long().cpu()])) < 5e-6
    assert rel_err(vv_rows, (v[col.long().cpu()] + (vr - bv)[eid.long().cpu()])) < 5e-6      # the bias enters in the row epilogue
    assert torch.equal(op('attn_modes_cached')(dev(q), kk, vv, rowptr, dev(bv), K, ops.LAYOUT_AGENT_MAJOR, col.numel()), got)


def test_op_mask_pairs_and_coalesce_max():
    g = torch.Generator().manual_seed(140)
    ms, md = torch.rand(5, 12, generator=g) > 0.3, torch.rand(5, 7, generator=g) > 0.3
    ei, off = op('mask_pairs')(ms.to(DEV), md.to(DEV), -(1 << 31), (1 << 31) - 1)
    b, i, j = (ms.unsqueeze(2) & md.unsqueeze(1)).nonzero(as_tuple=True)
    assert torch.equal(ei.cpu(), torch.stack([b * 12 + i, b * 7 + j]))
    assert int(off[-1]) == ei.size(1)
    e = torch.randint(0, 30, (2, 400), generator=g)
    attr = torch.randint(0, 5, (400,), generator=g).to(torch.uint8)
    out = op('coalesce_max')(e.to(DEV), attr.to(DEV), 30)
    key = e[0] * 30 + e[1]
    uniq, inv = torch.unique(key, return_inverse=True)
    want_attr = torch.zeros(uniq.numel(), dtype=torch.uint8).scatter_reduce(0, inv, attr, 'amax')
    assert torch.equal(out[0].cpu(), torch.stack([uniq // 30, uniq % 30]))
    assert torch.equal(out[1].cpu(), want_attr)


def test_op_frame_transforms_metrics_nms():
    g = torch.Generator().manual_seed(150)
    A, T, K = 9, 30, 6
    pos, head = torch.randn(A, T, 3, generator=g) * 20, (torch.rand(A, T, generator=g) * 2 - 1) * 3.0
    tgt = op('to_agent_frame')(pos.to(DEV), head.to(DEV), 9, 10, 20)
    th = head[:, 9]
    d = pos[:, 10:, :2] - pos[:, 9:10, :2]
    want_xy = torch.stack([d[..., 0] * th.cos().unsqueeze(1) + d[..., 1] * th.sin().unsqueeze(1),
                           -d[..., 0] * th.sin().unsqueeze(1) + d[..., 1] * th.cos().unsqueeze(1)], -1)
    assert torch.allclose(tgt[..., :2].cpu(), want_xy, atol=1e-4)
    traj = torch.randn(A, K, 20, 2, generator=g) * 5
    rows = torch.tensor([1, 4, 7])
    world = op('to_world_frame')(traj.to(DEV), pos.to(DEV), head.to(DEV), 9, rows.to(DEV)).cpu()
    c, s = th[rows].cos().view(-1, 1, 1), th[rows].sin().view(-1, 1, 1)
    t = traj[rows]
    want_w = torch.stack([t[..., 0] * c - t[..., 1] * s, t[..., 0] * s + t[..., 1] * c], -1) + pos[rows, 9, :2].view(-1, 1, 1, 2)
    assert torch.allclose(world, want_w, atol=1e-4)
    prob = torch.softmax(torch.randn(A, K, generator=g), -1)
    rec = op('motion_metrics')(traj.to(DEV), tgt[..., :2].contiguous(), prob.to(DEV), None, None, 2, ops.HEAD_NONE, 1, 6, True, 2.0).cpu()
    fde = (traj[:, :, -1] - tgt[:, -1, :2].cpu().unsqueeze(1)).norm(dim=-1).min(dim=1).values
    assert torch.allclose(rec[:, ops.METRIC_COLUMNS['fde']], fde, rtol=1e-5, atol=1e-5)
    assert bool((rec[:, ops.METRIC_COLUMNS['counted']] == 1).all())
    gt = tgt[..., :2].contiguous()
    sel, score = op('nms_select')(torch.randn(A, 12, 20, 2, generator=g).to(DEV), torch.rand(A, 12, generator=g).to(DEV), gt, 6, 6,
                                  2.5, 2.0)
    assert tuple(sel.shape) == (A, 2, 6) and tuple(score.shape) == (A, 2, 3)
    assert bool(((sel >= 0) & (sel < 12)).all())


def test_every_registered_op_is_tested_and_refuses_cpu_tensors():
    assert set(ops.OP_NAMES) == TESTED, set(ops.OP_NAMES) ^ TESTED
    from tempenslearnagg_qcnet_amd._lib import QcnError
    with pytest.raises(QcnError, match='GPU only'):
        Q.row_normalize(torch.randn(4, 128), 1e-5)


def test_ops_trace_under_fake_tensors():
    """The fake implementations give shapes / dtypes without launching anything (FakeTensor, torch.compile tracing)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(77, 128, device=DEV)
        wt = torch.empty(2 * (128 * 128 + 2), device=DEV)
        outs = Q.ln_linear(x, None, None, wt, None, True, ops.MATMUL_SPLIT_F16)
        assert [tuple(o.shape) for o in outs] == [(77, 128)] * 3
        rowptr, col, eid = Q.csr_build(torch.empty(2, 500, dtype=torch.int64, device=DEV), 77)
        assert tuple(rowptr.shape) == (78,) and rowptr.dtype == torch.int32 and tuple(col.shape) == (500,)
        pack = torch.empty(int(ops._lib.lib().qcn_fourier_pack_floats(3)), device=DEV)
        y = Q.fourier_embed(torch.empty(50, 3, device=DEV), 3, pack, None, 1, 3, 50, ops.MATMUL_BF16)
        assert y.dtype == torch.bfloat16 and tuple(y.shape) == (50, 128)
        out = Q.attn_fused(x, x, x, torch.empty(500, 128, device=DEV), rowptr, col, eid, torch.empty(128, 128, device=DEV),
                           torch.empty(128, 128, device=DEV), torch.empty(128, device=DEV), 1, 0, 0, None, None)
        assert tuple(out.shape) == (77, 128)
