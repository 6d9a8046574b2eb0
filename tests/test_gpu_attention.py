"""GPU parity: the fused attention kernel and AttentionLayer against the oracle (fp32, 1e-4 relative)."""
import pytest
import torch

from helpers import c_attention, cpu_state_dict, oracle, randomize_, rel_err

from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.layers import AttentionLayer
from tempenslearnagg_qcnet_amd.utils.graph import CSRGraph

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture
def restore_matmul_mode():
    mode = ops.matmul_mode()
    yield
    ops.set_matmul_mode(mode)
TOL = 1e-4   # BASELINE.json north_star: fp32 outputs within 1e-4 relative


def _graph(n_src, n_dst, E, seed, heavy=None, empty_from=None):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n_src, (E,), generator=g)
    hi = n_dst if empty_from is None else empty_from
    dst = torch.randint(0, hi, (E,), generator=g)
    if heavy is not None:
        dst[:heavy] = 3
    return torch.stack([src, dst])


def _feat(n, seed, scale=1.0):
    return torch.randn(n, 128, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize('n_src,n_dst,E,has_r,heavy,empty_from', [
    (40, 30, 500, True, None, None),
    (40, 30, 500, False, None, None),
    (300, 257, 9000, True, 700, None),       # one destination with 700+ in-edges, n_dst not a multiple of 4
    (64, 100, 333, True, None, 60),          # destinations 60.. have no edges -> exact zeros
    (5, 9, 0, True, None, None),             # no edges at all
    (1, 1, 3, False, None, None),
])
def test_attn_kernel_vs_fp64_oracle(n_src, n_dst, E, has_r, heavy, empty_from):
    ei = _graph(n_src, n_dst, E, 11, heavy, empty_from)
    q, k, v = _feat(n_dst, 1, 1.5), _feat(n_src, 2, 1.5), _feat(n_src, 3)
    kr = _feat(E, 4, 0.5) if has_r else None
    vr = _feat(E, 5, 0.5) if has_r else None
    want = c_attention(q, k, v, kr, vr, ei)
    graph = CSRGraph.from_edge_index(ei.to(DEV), n_src, n_dst)
    got = ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), None if kr is None else kr.to(DEV),
                       None if vr is None else vr.to(DEV), graph.rowptr, graph.col, graph.eid)
    torch.cuda.synchronize()
    assert got.shape == (n_dst, 128)
    if E:
        assert rel_err(got, want) < 1e-5
    deg = torch.bincount(ei[1], minlength=n_dst)
    assert bool((got.cpu()[deg == 0] == 0).all())


def test_attn_kernel_large_logits_and_edge_permutation_invariance():
    n_src, n_dst, E = 200, 150, 6000
    ei = _graph(n_src, n_dst, E, 12)
    q, k, v = _feat(n_dst, 1, 6.0), _feat(n_src, 2, 6.0), _feat(n_src, 3)   # logits of +-100: softmax must not overflow
    kr, vr = _feat(E, 4), _feat(E, 5)
    want = c_attention(q, k, v, kr, vr, ei)
    def run(perm):
        e = ei[:, perm].to(DEV)
        g = CSRGraph.from_edge_index(e, n_src, n_dst)
        return ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), kr[perm].to(DEV), vr[perm].to(DEV), g.rowptr, g.col, g.eid)
    a = run(torch.arange(E))
    b = run(torch.randperm(E, generator=torch.Generator().manual_seed(1)))
    assert torch.isfinite(a).all()
    assert rel_err(a, want) < 1e-5 and rel_err(b, want) < 1e-5


def test_attn_wrapper_rejects_bad_inputs():
    ei = _graph(10, 10, 50, 13).to(DEV)
    g = CSRGraph.from_edge_index(ei, 10, 10)
    q = torch.randn(10, 128, device=DEV)
    with pytest.raises(Exception):
        ops.attn_fwd(q.cpu(), q, q, None, None, g.rowptr, g.col, g.eid)           # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.attn_fwd(q[:, :64].contiguous(), q, q, None, None, g.rowptr, g.col, g.eid)
    with pytest.raises(ValueError):
        ops.attn_fwd(q, q, q, torch.randn(49, 128, device=DEV), torch.randn(49, 128, device=DEV), g.rowptr, g.col, g.eid)
    with pytest.raises(IndexError):
        CSRGraph.from_edge_index(ei + 5, 10, 10)


@pytest.mark.parametrize('bipartite,has_pos_emb', [(True, True), (False, True), (False, False)])
def test_attention_layer_vs_oracle(bipartite, has_pos_emb):
    torch.manual_seed(3)
    layer = randomize_(AttentionLayer(hidden_dim=128, num_heads=8, head_dim=16, dropout=0.1, bipartite=bipartite,
                                      has_pos_emb=has_pos_emb), seed=5).eval()
    sd = cpu_state_dict(layer)
    n_src, n_dst, E = (90, 70, 2500) if bipartite else (80, 80, 2500)
    ei = _graph(n_src, n_dst, E, 21, heavy=300, empty_from=n_dst - 7)
    x_src, x_dst = _feat(n_src, 6), _feat(n_dst, 7)
    r = _feat(E, 8) if has_pos_emb else None
    x_cpu = (x_src, x_dst) if bipartite else x_dst
    with torch.no_grad():
        want = oracle.attention_layer(sd, '', x_cpu, r, ei)
        layer = layer.to(DEV)
        x_gpu = (x_src.to(DEV), x_dst.to(DEV)) if bipartite else x_dst.to(DEV)
        got = layer(x_gpu, None if r is None else r.to(DEV), ei.to(DEV))
        # a prebuilt CSRGraph gives the same result as the raw edge_index
        got2 = layer(x_gpu, None if r is None else r.to(DEV), CSRGraph.from_edge_index(ei.to(DEV), n_src, n_dst))
    assert rel_err(got, want) < TOL
    assert torch.equal(got, got2)


def test_attention_layer_state_dict_keys_match_reference_names():
    layer = AttentionLayer(hidden_dim=128, num_heads=8, head_dim=16, dropout=0.1, bipartite=False, has_pos_emb=True)
    keys = set(layer.state_dict().keys())
    # aliased prenorm of non-bipartite layers appears under both names
    for k in ('to_q.weight', 'to_q.bias', 'to_k.weight', 'to_v.bias', 'to_k_r.weight', 'to_v_r.bias', 'to_s.weight',
              'to_g.weight', 'to_out.bias', 'ff_mlp.0.weight', 'ff_mlp.3.bias', 'attn_prenorm_x_src.weight',
              'attn_prenorm_x_dst.weight', 'attn_prenorm_r.bias', 'attn_postnorm.weight', 'ff_prenorm.bias',
              'ff_postnorm.weight'):
        assert k in keys, k
    assert 'to_k.bias' not in keys and 'to_k_r.bias' not in keys
    assert layer.to_g.weight.shape == (128, 256) and layer.ff_mlp[0].weight.shape == (512, 128)
    assert layer.attn_prenorm_x_dst is layer.attn_prenorm_x_src
    assert len(list(layer.parameters())) == len(keys) - 2


# ---------------------------------------------------------------------------------------------------------
# fused positional form (qcn_attn_fused_f32) against the fp64 oracle fed with materialised kr / vr
# ---------------------------------------------------------------------------------------------------------
def _pos_weights(seed):
    g = torch.Generator().manual_seed(seed)
    w_kr = torch.randn(128, 128, generator=g) * 0.15
    w_vr = torch.randn(128, 128, generator=g) * 0.15
    b_vr = torch.randn(128, generator=g) * 0.1
    gamma = 1.0 + torch.randn(128, generator=g) * 0.1
    beta = torch.randn(128, generator=g) * 0.1
    return w_kr, w_vr, b_vr, gamma, beta


def _materialise(r, w_kr, w_vr, b_vr, gamma, beta):
    rn = torch.nn.functional.layer_norm(r.double(), (128,), gamma.double(), beta.double(), 1e-5)
    return (rn @ w_kr.double().t()).float(), (rn @ w_vr.double().t() + b_vr.double()).float()


def _fused(q, k, v, r, graph, w_kr, w_vr, b_vr, gamma, beta, rep=1, layout=0):
    wk = (w_kr * gamma[None]).to(DEV).contiguous()
    wv = (w_vr * gamma[None]).to(DEV).contiguous()
    bv = (w_vr @ beta + b_vr).to(DEV).contiguous()
    rhat = ops.row_normalize(r.to(DEV))
    return ops.attn_fused(q.to(DEV), k.to(DEV), v.to(DEV), rhat, graph.rowptr, graph.col, graph.eid, wk, wv, bv, rep, layout)


@pytest.mark.parametrize('n_src,n_dst,E,heavy,empty_from', [
    (40, 30, 500, None, None),
    (300, 257, 9000, 700, None),
    (64, 100, 333, None, 60),
    (5, 9, 0, None, None),
    (700, 5000, 60000, None, None),          # more destinations than one pass of 256 workgroups x 16 waves
])
def test_attn_fused_vs_fp64_oracle(n_src, n_dst, E, heavy, empty_from):
    ei = _graph(n_src, n_dst, E, 31, heavy, empty_from)
    q, k, v, r = _feat(n_dst, 1, 1.5), _feat(n_src, 2, 1.5), _feat(n_src, 3), _feat(E, 4, 2.0) + 0.3
    pw = _pos_weights(9)
    kr, vr = _materialise(r, *pw)
    want = c_attention(q, k, v, kr if E else None, vr if E else None, ei)
    graph = CSRGraph.from_edge_index(ei.to(DEV), n_src, n_dst)
    got = _fused(q, k, v, r, graph, *pw)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    if E:
        assert rel_err(got, want) < 2e-5
    deg = torch.bincount(ei[1], minlength=n_dst)
    assert bool((got.cpu()[deg == 0] == 0).all())


def test_row_normalize_matches_layer_norm():
    x = _feat(1001, 5, 3.0) + 1.0
    got = ops.row_normalize(x.to(DEV))
    want = torch.nn.functional.layer_norm(x.double(), (128,), None, None, 1e-5)
    assert rel_err(got, want) < 1e-6


@pytest.mark.parametrize('layout', [1, 2])
def test_attn_fused_mode_replication_equals_explicit_graph(layout):
    """K mode queries per row sharing one edge list (qcnet_decoder.py:170,190-192) == the K-times replicated graph."""
    K, n_rows, n_src, E = 6, 37, 50, 900
    base = _graph(n_src, n_rows, E, 41, heavy=100, empty_from=30)
    q, k, v, r = _feat(n_rows * K, 1, 1.5), _feat(n_src, 2, 1.5), _feat(n_src, 3), _feat(E, 4, 2.0)
    pw = _pos_weights(10)
    if layout == 1:      # mode-major destinations: node = m * n_rows + row; r.repeat(K, 1)
        full = torch.cat([torch.stack([base[0], base[1] + m * n_rows]) for m in range(K)], dim=1)
        r_full = r.repeat(K, 1)
    else:                # agent-major destinations: node = row * K + m; r.repeat_interleave(K)
        full = torch.stack([base[0].repeat_interleave(K), (base[1][:, None] * K + torch.arange(K)[None]).reshape(-1)])
        r_full = r.repeat_interleave(K, dim=0)
    kr, vr = _materialise(r_full, *pw)
    want = c_attention(q, k, v, kr, vr, full)
    g = CSRGraph.from_edge_index(base.to(DEV), n_src, n_rows).replicate(K, layout)
    got = _fused(q, k, v, r, g, *pw, rep=K, layout=layout)
    assert rel_err(got, want) < 2e-5


def _modes(q, k, v, r, graph, w_kr, w_vr, b_vr, gamma, beta, rep, layout):
    wpack = ops.pack_attn_modes((w_kr * gamma[None]).to(DEV), (w_vr * gamma[None]).to(DEV))
    bv = (w_vr @ beta + b_vr).to(DEV).contiguous()
    rs = ops.split_rows(ops.row_normalize(r.to(DEV)))
    return ops.attn_modes(q.to(DEV), k.to(DEV), v.to(DEV), rs, graph.rowptr, graph.col, graph.eid, wpack, bv, rep, layout)


@pytest.mark.parametrize('layout,K,n_rows,n_src,E,heavy,empty_from', [
    (2, 6, 37, 50, 900, 100, 30),
    (1, 6, 37, 50, 900, 100, 30),
    (2, 6, 3000, 400, 90000, 300, None),     # more rows than workgroups, rows of 1..300+ edges (many tiles per row)
    (2, 4, 64, 20, 700, None, None),
    (2, 6, 5, 9, 0, None, None),
])
@pytest.mark.parametrize('lean', ['0', '1'])
def test_attn_modes_vs_fp64_oracle(layout, K, n_rows, n_src, E, heavy, empty_from, lean, monkeypatch):
    monkeypatch.setenv('QCN_MODES_LEAN', lean)       # both kernel variants (pipelined one-per-CU, lean two-per-CU)
    _attn_modes_case(layout, K, n_rows, n_src, E, heavy, empty_from)


def test_attn_modes_on_a_rowptr_slice():
    """A rowptr that is a slice of a larger CSR (first entry > 0, col / eid / rhat indexed by absolute edge ids): the
    edge-balanced row chunks (attention_modes.hip: row_chunk) must still cover exactly the slice's rows."""
    K, n_rows, n_src, E = 6, 700, 120, 30000
    base = _graph(n_src, n_rows, E, 47, heavy=200)
    q, k, v, r = _feat(n_rows * K, 5, 1.5), _feat(n_src, 6, 1.5), _feat(n_src, 7), _feat(E, 8, 2.0)
    pw = _pos_weights(13)
    g = CSRGraph.from_edge_index(base.to(DEV), n_src, n_rows).replicate(K, 2)
    full = _modes(q, k, v, r, g, *pw, rep=K, layout=2)
    w_kr, w_vr, b_vr, gamma, beta = pw
    wpack = ops.pack_attn_modes((w_kr * gamma[None]).to(DEV), (w_vr * gamma[None]).to(DEV))
    bv = (w_vr @ beta + b_vr).to(DEV).contiguous()
    rs = ops.split_rows(ops.row_normalize(r.to(DEV)))
    for a, b in ((250, 700), (1, 2), (333, 400)):
        rowptr = g.rowptr[a:b + 1].contiguous()
        assert int(rowptr[0]) > 0
        part = ops.attn_modes(q[a * K:b * K].to(DEV).contiguous(), k.to(DEV), v.to(DEV), rs, rowptr, g.col, g.eid, wpack, bv, K, 2)
        assert torch.equal(part, full[a * K:b * K])


def _attn_modes_case(layout, K, n_rows, n_src, E, heavy, empty_from):
    """qcn_attn_modes_f32 (mode-replicated relations on the matrix cores) against the fp64 oracle on the explicitly
    K-times replicated graph, and against qcn_attn_fused_f32 on the same inputs.  Tolerance 2e-5 relative."""
    base = _graph(n_src, n_rows, E, 43, heavy=heavy, empty_from=empty_from)
    q, k, v, r = _feat(n_rows * K, 1, 1.5), _feat(n_src, 2, 1.5), _feat(n_src, 3), _feat(E, 4, 2.0)
    pw = _pos_weights(12)
    g = CSRGraph.from_edge_index(base.to(DEV), n_src, n_rows).replicate(K, layout)
    got = _modes(q, k, v, r, g, *pw, rep=K, layout=layout)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    if E == 0:
        assert bool((got == 0).all())
        return
    if layout == 1:
        full = torch.cat([torch.stack([base[0], base[1] + m * n_rows]) for m in range(K)], dim=1)
        r_full = r.repeat(K, 1)
    else:
        full = torch.stack([base[0].repeat_interleave(K), (base[1][:, None] * K + torch.arange(K)[None]).reshape(-1)])
        r_full = r.repeat_interleave(K, dim=0)
    kr, vr = _materialise(r_full, *pw)
    want = c_attention(q, k, v, kr, vr, full)
    assert rel_err(got, want) < 2e-5
    ref = _fused(q, k, v, r, g, *pw, rep=K, layout=layout)
    assert rel_err(got, ref) < 2e-5
    deg = torch.bincount(full[1], minlength=n_rows * K)
    assert bool((got.cpu()[deg == 0] == 0).all())


@pytest.mark.parametrize('layout,K,n_rows,n_src,E,heavy,empty_from,sorted_', [
    (2, 6, 37, 50, 900, 100, 30, False),
    (1, 6, 37, 50, 900, 100, 30, False),
    (2, 6, 3000, 400, 90000, 300, None, False),   # more rows than workgroups, rows of 1 .. 300+ edges
    (2, 3, 700, 90, 20000, None, None, True),     # no edge ids (COO order = CSR order), rep < 4
    (2, 6, 5, 9, 0, None, None, False),
])
@pytest.mark.parametrize('bf16', [False, True])
def test_attn_modes_store_and_cached_equal_recomputation(layout, K, n_rows, n_src, E, heavy, empty_from, sorted_, bf16):
    """qcn_attn_modes_store_* returns what qcn_attn_modes_* returns and the per-edge projections kk_e = k[j] + W_kr LN_r(r_e),
    vv_e = v[j] + W_vr LN_r(r_e) in CSR-slot order (checked against an fp64 form); qcn_attn_modes_cached_f32 with those tables
    and ANOTHER q is bit-identical to qcn_attn_modes_* recomputing everything for that q (the decoder's recurrent proposal
    steps: same layer, same sources, new mode queries)."""
    base = _graph(n_src, n_rows, E, 43, heavy=heavy, empty_from=empty_from)
    q, k, v, r = _feat(n_rows * K, 1, 1.5), _feat(n_src, 2, 1.5), _feat(n_src, 3), _feat(E, 4, 2.0)
    q2 = _feat(n_rows * K, 21, 1.5)
    w_kr, w_vr, b_vr, gamma, beta = _pos_weights(12)
    g = CSRGraph.from_edge_index(base.to(DEV), n_src, n_rows)
    if sorted_:      # edge ids dropped: feature rows in CSR order
        r = r[g.eid.long().cpu()]
        g = CSRGraph(None, n_src, n_rows, g.rowptr, g.col, None)
    g = g.replicate(K, layout)
    mode = ops.MATMUL_BF16 if bf16 else ops.MATMUL_SPLIT_F16
    wpack = ops.pack_attn_modes((w_kr * gamma[None]).to(DEV), (w_vr * gamma[None]).to(DEV), mode=mode)
    bv = (w_vr @ beta + b_vr).to(DEV).contiguous()
    rhat = ops.row_normalize(r.to(DEV)) if E else torch.zeros(0, 128, device=DEV)
    rs = rhat.bfloat16() if bf16 else ops.split_rows(rhat)
    args = (k.to(DEV), v.to(DEV), rs, g.rowptr, g.col, g.eid, wpack, bv, K, layout)
    plain = ops.attn_modes(q.to(DEV), *args)
    out, kk, vv = ops.attn_modes_store(q.to(DEV), *args)
    assert torch.equal(out, plain)
    assert kk.numel() == vv.numel() == 128 * ops.attn_modes_table_slots(E, n_rows)
    if E:
        kk_rows, vv_rows = ops.attn_modes_table_rows(kk, vv, g.rowptr)     # head-major tables -> [E, 128] rows in CSR order
        rid = (g.eid if g.eid is not None else torch.arange(E, device=DEV)).long()
        wk64, wv64 = (w_kr * gamma[None]).double().to(DEV), (w_vr * gamma[None]).double().to(DEV)
        rh = rhat[rid].bfloat16().double() if bf16 else rhat[rid].double()
        if bf16:
            wk64, wv64 = wk64.float().bfloat16().double(), wv64.float().bfloat16().double()
        want_kk = k.to(DEV).double()[g.col.long()] + rh @ wk64.t()
        want_vv = v.to(DEV).double()[g.col.long()] + rh @ wv64.t()
        assert kk_rows.shape == (E, 128) and rel_err(kk_rows, want_kk) < 5e-6 and rel_err(vv_rows, want_vv) < 5e-6
    again = ops.attn_modes_cached(q2.to(DEV), kk, vv, g.rowptr, bv, K, layout, E)
    assert torch.equal(again, ops.attn_modes(q2.to(DEV), *args))
    assert torch.equal(ops.attn_modes_cached(q.to(DEV), kk, vv, g.rowptr, bv, K, layout), plain)


@pytest.mark.parametrize('mode,h0_scale', [(0, 0.3), (1, 0.3), (1, 5.0)])
def test_gru_kernel_matches_torch_gru(mode, h0_scale, restore_matmul_mode):
    """qcn_gru_f32 + input GEMM against torch.nn.GRU (CPU fp32), ragged sequence count (tail block), in both matmul
    modes; h0_scale 5.0 puts the initial state outside [-1, 1] (the split kernel's fixed state scale must follow it)."""
    ops.set_matmul_mode(mode)
    torch.manual_seed(11)
    gru = torch.nn.GRU(128, 128).eval()
    with torch.no_grad():
        for p in gru.parameters():
            p.copy_(torch.randn_like(p) * 0.12)
    n_seq, n_steps = 77, 60
    seq = torch.randn(n_steps, n_seq, 128)
    h0 = torch.randn(128) * h0_scale
    with torch.no_grad():
        want = gru(seq, h0.view(1, 1, 128).repeat(1, n_seq, 1))[1].squeeze(0)
    x = seq.transpose(0, 1).reshape(-1, 128).contiguous().to(DEV)          # row = seq * n_steps + t
    wih_t = ops.pack_chain_matrices(list(gru.weight_ih_l0.detach().view(3, 128, 128).to(DEV)))
    got = ops.gru_last_hidden(x, n_steps, wih_t, gru.bias_ih_l0.detach().view(3, 128).contiguous().to(DEV),
                              ops.pack_gru_whh(gru.weight_hh_l0.detach().to(DEV), h0.to(DEV)), gru.bias_hh_l0.detach().to(DEV),
                              h0.to(DEV))
    assert rel_err(got, want) < 1e-5


@pytest.mark.gpu
def test_split_matmul_error(restore_matmul_mode):
    """The split-f16 matmul mode (include/qcnet_hip.h: QCN_MATMUL_SPLIT_F16) against fp64, next to the exact fp32 MFMA
    mode on the same inputs.  Error is measured per output against its own scale sum_k |x_k| |w_k| (the backward-error
    scale of a dot product); the bar is 2^-19 for both modes — a few fp32 ulps, two orders below the 1e-4 parity budget
    (measured: exact 1.4e-6 — 128 sequential fp32 roundings —, split 8.7e-7).
    Rows span 35 decades of magnitude (the per-row power-of-two scaling), one row is zero, the four matrices differ in
    magnitude by 10^6 (the per-matrix scaling)."""
    torch.manual_seed(11)
    n = 3000
    x = torch.randn(n, 128, device=DEV) * (10.0 ** torch.empty(n, 1, device=DEV).uniform_(-20, 15))
    x[7] = 0.0
    x[8, 3:] = 0.0                                  # a single non-zero element
    ws = [torch.randn(128, 128, device=DEV) * s for s in (1e-3, 0.05, 1.0, 300.0)]
    bias = torch.randn(4, 128, device=DEV)
    errs = {}
    for mode in (ops.MATMUL_EXACT_F32, ops.MATMUL_SPLIT_F16):
        ops.set_matmul_mode(mode)
        assert ops.matmul_mode() == mode
        ys, _ = ops.ln_linear(x, None, None, ops.pack_chain_matrices(ws), bias)
        worst = 0.0
        for j, w in enumerate(ws):
            ref = x.double() @ w.double().t() + bias[j].double()
            scale = x.double().abs() @ w.double().abs().t() + bias[j].double().abs()
            worst = max(worst, float(((ys[j].double() - ref).abs() / scale.clamp_min(1e-300)).max()))
            assert torch.isfinite(ys[j]).all()
        errs[mode] = worst
    print('matmul backward error: exact %.3g  split %.3g' % (errs[ops.MATMUL_EXACT_F32], errs[ops.MATMUL_SPLIT_F16]))
    assert errs[ops.MATMUL_EXACT_F32] < 2.0 ** -19
    assert errs[ops.MATMUL_SPLIT_F16] < 2.0 ** -19


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [ops.MATMUL_EXACT_F32, ops.MATMUL_SPLIT_F16])
def test_attention_layer_vs_oracle_in_both_matmul_modes(mode, restore_matmul_mode):
    ops.set_matmul_mode(mode)
    test_attention_layer_vs_oracle(True, True)
    test_attention_layer_vs_oracle(False, True)


@pytest.mark.parametrize('n_seq,in_dim,folded,mode', [(77, 2, True, 1), (64, 2, False, 1), (1, 3, True, 1), (300, 2, False, 2),
                                                      (33, 3, True, 2)])
def test_trajectory_gru_blocked_layout_equals_row_major(n_seq, in_dim, folded, mode, restore_matmul_mode):
    """ops.trajectory_gru (qcn_fourier_gru_input_f32 + qcn_gru_blocked_f32: the GRU's input halves in the recurrence's own
    blocked layout, a tile = one step of 32 sequences) against ops.fourier_linear + ops.gru_recurrence (row-major input halves)
    — the same input halves bit for bit (the two chain kernels share their arithmetic); the blocked recurrence is the two-group
    kernel, whose gates fold log2(e) into their constants and carry the state pre-scaled, so the 60-step results agree to a
    few ulps of the state's range, not bit for bit: |diff| <= 2e-6 on values in [-1, 1].  Ragged sequence counts (padding of
    the last block, an odd number of blocks), one sequence, both tail forms, split and bf16 modes."""
    from tempenslearnagg_qcnet_amd.layers import FourierEmbedding
    ops.set_matmul_mode(mode)
    torch.manual_seed(5)
    n_steps = 60
    emb = FourierEmbedding(input_dim=in_dim, hidden_dim=128, num_freq_bands=64).eval().to(DEV)
    with torch.no_grad():
        for name, p in emb.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    gru = torch.nn.GRU(128, 128).eval().to(DEV)
    with torch.no_grad():
        for p in gru.parameters():
            p.copy_(torch.randn_like(p) * 0.12)
    h0 = torch.randn(128, device=DEV) * 0.3
    x = torch.randn(n_seq * n_steps, in_dim, device=DEV) * 20.0
    ws, bs = list(gru.weight_ih_l0.detach().view(3, 128, 128)), list(gru.bias_ih_l0.detach().view(3, 128))
    with torch.no_grad():
        if folded:
            lin = emb.to_out[2]
            ws, bs = ops.fold_linear(ws, bs, lin.weight, lin.bias)
        else:
            bs = torch.stack(bs).contiguous()
        wt = ops.pack_chain_matrices(ws, mode=mode)
        whh = ops.pack_gru_whh(gru.weight_hh_l0.detach(), h0, mode=mode)
        b_hh = gru.bias_hh_l0.detach().contiguous()
        gi = ops.fourier_linear(x, in_dim, emb.packed_weights(), wt, bs, folded=folded, mode=mode)
        want = ops.gru_recurrence(gi[0], gi[1], gi[2], n_steps, whh, b_hh, h0, mode=mode)
        got = ops.trajectory_gru(x, in_dim, n_steps, emb.packed_weights(), wt, bs, folded, whh, b_hh, h0, mode=mode)
    assert got.shape == (n_seq, 128) and bool(torch.isfinite(got).all())
    assert float((got - want).abs().max()) <= 2e-6


@pytest.mark.parametrize('mode', [1, 0])
def test_chained_layers_equal_unchained(mode, restore_matmul_mode):
    """A stack bipartite -> one-node-set -> bipartite run chained (each layer's node-side kernel applies the next layer's
    destination-side LayerNorm + projections and computes to_s itself: qcn_attn_post_chain_f32) against the same stack with
    every layer projecting its own rows (qcn_ln_linear_f32 + qcn_attn_post_f32): the same arithmetic per element, up to
    the order in which a bias joins its dot product (1e-5)."""
    from tempenslearnagg_qcnet_amd.layers import AttentionLayer, attention_layer
    ops.set_matmul_mode(mode)
    torch.manual_seed(31)
    n_src, n_dst, E = 90, 333, 4000
    layers = [AttentionLayer(128, 8, 16, 0.0, bipartite=b, has_pos_emb=True).to(DEV).eval() for b in (True, False, True)]
    with torch.no_grad():
        for layer in layers:
            for name, p in layer.named_parameters():
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.1)
    x_src, x = torch.randn(n_src, 128, device=DEV), torch.randn(n_dst, 128, device=DEV)
    g = torch.Generator().manual_seed(5)
    ei_b = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst, (E,), generator=g)]).to(DEV)
    ei_s = torch.stack([torch.randint(0, n_dst, (E,), generator=g), torch.randint(0, n_dst, (E,), generator=g)]).to(DEV)
    r_b, r_s = torch.randn(E, 128, device=DEV), torch.randn(E, 128, device=DEV)

    def run(chain):
        attention_layer.CHAIN_NODE_SIDE = chain
        with torch.no_grad():
            if not chain:
                y = layers[0]((x_src, x), r_b, ei_b)
                y = layers[1](y, r_s, ei_s)
                return layers[2]((x_src, y), r_b, ei_b)
            y, pre = layers[0]((x_src, x), r_b, ei_b, next_layer=layers[1])
            assert len(pre) == 3
            y, pre = layers[1](y, r_s, ei_s, pre=pre, next_layer=layers[2])
            assert len(pre) == 1
            return layers[2]((x_src, y), r_b, ei_b, pre=pre)
    try:
        want, got = run(False), run(True)
    finally:
        attention_layer.CHAIN_NODE_SIDE = True
    assert rel_err(got, want) < 1e-5
