"""GPU parity: edge construction is bit-exact against the oracle (BASELINE.json: "bit-exact edge_index")."""
import pytest
import torch

from helpers import c_radius, oracle

import tempenslearnagg_qcnet_amd as pkg
from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.utils import graph as G

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _points(n, extent, seed, dim=2):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, dim, generator=g) * 2 - 1) * extent


def _ptr(sizes):
    p = torch.zeros(len(sizes) + 1, dtype=torch.int64)
    p[1:] = torch.tensor(sizes).cumsum(0)
    return p


@pytest.mark.parametrize('sizes_x,sizes_y,r,cap,dim', [
    ([50], [150], 50.0, 300, 2),
    ([50, 37, 0, 64], [150, 90, 10, 0], 60.0, 300, 2),          # ragged + empty examples
    ([200, 180], [400, 350], 100.0, 32, 2),                       # cap truncation (lowest x indices survive)
    ([33, 21], [40, 17], 70.0, 8, 3),
])
def test_radius_bit_exact(sizes_x, sizes_y, r, cap, dim):
    x = _points(sum(sizes_x), 100.0, 1, dim)
    y = _points(sum(sizes_y), 100.0, 2, dim)
    px, py = _ptr(sizes_x), _ptr(sizes_y)
    want = c_radius(x, y, r, px, py, cap)
    got, off = ops.radius_pairs(x.to(DEV), y.to(DEV), r, px.to(DEV), py.to(DEV), cap)
    assert got.dtype == torch.int64
    assert torch.equal(got.cpu(), want)
    # offset is the CSR pointer over y
    counts = torch.bincount(want[0], minlength=y.size(0))
    assert torch.equal(off.cpu()[1:] - off.cpu()[:-1], counts)
    # the torch restatement agrees with the C restatement (pins the two oracle forms to each other)
    bx = torch.repeat_interleave(torch.arange(len(sizes_x)), torch.tensor(sizes_x))
    by = torch.repeat_interleave(torch.arange(len(sizes_y)), torch.tensor(sizes_y))
    if len(sizes_x) > 1 and sizes_x[-1] > 0 and sizes_y[-1] > 0:
        assert torch.equal(oracle.radius(x, y, r, bx, by, cap), want)


def test_radius_on_boundary_distances():
    # integer lattice: many pairs sit exactly on d^2 == r^2 (strict '<' must drop them)
    xs = torch.stack(torch.meshgrid(torch.arange(-6., 7.), torch.arange(-6., 7.), indexing='ij'), -1).reshape(-1, 2)
    p = _ptr([xs.size(0)])
    want = c_radius(xs, xs, 5.0, p, p, 1000)
    got, _ = ops.radius_pairs(xs.to(DEV), xs.to(DEV), 5.0, p.to(DEV), p.to(DEV), 1000)
    assert torch.equal(got.cpu(), want)
    d2 = ((xs[want[0]] - xs[want[1]]) ** 2).sum(-1)
    assert float(d2.max()) < 25.0


def test_radius_api_matches_oracle_batched():
    x, y = _points(300, 80.0, 3), _points(500, 80.0, 4)
    bx = torch.repeat_interleave(torch.arange(4), torch.tensor([70, 80, 90, 60]))
    by = torch.repeat_interleave(torch.arange(4), torch.tensor([100, 150, 120, 130]))
    want = oracle.radius(x, y, 40.0, bx, by, 300)
    got = G.radius(x.to(DEV), y.to(DEV), 40.0, bx.to(DEV), by.to(DEV), max_num_neighbors=300)
    assert torch.equal(got.cpu(), want)
    want_g = oracle.radius_graph(x, 40.0, bx, loop=False, max_num_neighbors=300)
    got_g = G.radius_graph(x.to(DEV), 40.0, bx.to(DEV), loop=False, max_num_neighbors=300)
    assert torch.equal(got_g.cpu(), want_g)
    # unbatched, with the cap biting: a target keeps 'cap' others only if its own index is among its first cap+1 hits
    want_c = oracle.radius_graph(x, 60.0, None, loop=False, max_num_neighbors=16)
    got_c = G.radius_graph(x.to(DEV), 60.0, None, loop=False, max_num_neighbors=16)
    assert torch.equal(got_c.cpu(), want_c)
    deg = torch.bincount(want_c[1], minlength=300)
    assert int(deg.max()) == 17 and int(deg.min()) >= 16   # the G2 quirk: 16 or 17 survivors


def test_radius_fused_filters_equal_post_filtering():
    x = _points(400, 60.0, 5)
    p = _ptr([150, 250])
    g = torch.Generator().manual_seed(6)
    mask = torch.rand(400, generator=g) > 0.3
    base = oracle.radius_graph(x, 25.0, torch.repeat_interleave(torch.arange(2), torch.tensor([150, 250])),
                               loop=False, max_num_neighbors=20)
    want = oracle.subgraph(mask, base)
    got, off = ops.radius_pairs(x.to(DEV), x.to(DEV), 25.0, p.to(DEV), p.to(DEV), 21, drop_self=True,
                                mask_x=mask.to(DEV), mask_y=mask.to(DEV), y_first=False)
    assert torch.equal(got.cpu(), want)
    assert int(off[-1]) == want.size(1)


def test_radius_empty_inputs():
    e = G.radius(torch.empty(0, 2, device=DEV), torch.rand(5, 2, device=DEV), 1.0)
    assert e.shape == (2, 0) and e.dtype == torch.int64
    x = torch.rand(5, 2, device=DEV) * 100
    e2 = G.radius(x, x + 1000.0, 1.0)
    assert e2.shape == (2, 0)


def test_csr_build_matches_stable_sort():
    g = torch.Generator().manual_seed(7)
    n_src, n_dst, E = 1000, 777, 50000
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst, (E,), generator=g)])
    ei[1, :100] = 5                                            # a heavy destination
    rowptr, col, eid = ops.csr_build(ei.to(DEV), n_dst)
    order = torch.sort(ei[1], stable=True)[1]
    assert torch.equal(eid.cpu().long(), order)
    assert torch.equal(col.cpu().long(), ei[0][order])
    want_ptr = torch.zeros(n_dst + 1, dtype=torch.long)
    want_ptr[1:] = torch.bincount(ei[1], minlength=n_dst).cumsum(0)
    assert torch.equal(rowptr.cpu().long(), want_ptr)
    # empty edge list
    rp, c, e = ops.csr_build(torch.empty(2, 0, dtype=torch.long, device=DEV), 10)
    assert torch.equal(rp.cpu(), torch.zeros(11, dtype=torch.int32)) and c.numel() == 0


def test_dense_to_sparse_and_coalesce_match_oracle():
    g = torch.Generator().manual_seed(8)
    adj = torch.rand(5, 7, 7, generator=g) > 0.6
    assert torch.equal(G.dense_to_sparse(adj.to(DEV))[0].cpu(), oracle.dense_to_sparse(adj))
    adj2 = torch.rand(5, 7, 3, generator=g) > 0.5
    assert torch.equal(G.bipartite_dense_to_sparse(adj2.to(DEV)).cpu(), oracle.bipartite_dense_to_sparse(adj2))
    ei = torch.randint(0, 30, (2, 400), generator=g)
    attr = torch.randint(0, 5, (400,), generator=g).to(torch.uint8)
    wi, wa = oracle.coalesce_max(ei, attr)
    gi, ga = G.merge_edges([ei[:, :150].to(DEV), ei[:, 150:].to(DEV)], [attr[:150].to(DEV), attr[150:].to(DEV)],
                           reduce='max')
    assert torch.equal(gi.cpu(), wi) and torch.equal(ga.cpu(), wa)
    # lexicographically sorted and unique
    key = wi[0] * 30 + wi[1]
    assert bool((key[1:] > key[:-1]).all())


@pytest.mark.gpu
def test_rel_geometry_kernel_matches_reference_helpers():
    """qcn_rel_geometry_f32 against the torch form built from the reference-named helpers (utils/geometry.py), including
    zero displacement (the signed-zero case: atan2 must give 0, not pi), angles on the wrap boundary and the optional
    fourth column."""
    import math
    from tempenslearnagg_qcnet_amd import ops
    from tempenslearnagg_qcnet_amd.modules._common import rel_geometry_torch
    torch.manual_seed(5)
    ns, nd, e = 700, 300, 20000
    pos_s = (torch.randn(ns, 3, device='cuda:0') * 40).contiguous()
    pos_d = (torch.randn(nd, 2, device='cuda:0') * 40).contiguous()
    head_s = torch.empty(ns, device='cuda:0').uniform_(-math.pi, math.pi)
    head_d = torch.empty(nd, device='cuda:0').uniform_(-math.pi, math.pi)
    head_d[:4] = torch.tensor([0.0, math.pi, -math.pi, math.pi / 2], device='cuda:0')
    ei = torch.stack([torch.randint(0, ns, (e,), device='cuda:0'), torch.randint(0, nd, (e,), device='cuda:0')])
    ei[0, :50] = torch.arange(50, device='cuda:0')                  # distinct sources for the
    pos_s[ei[0, :50], :2] = pos_d[ei[1, :50]]                     # zero-displacement edges
    head_s[ei[0, 50:80]] = head_d[ei[1, 50:80]]                    # zero heading difference
    extra = torch.randint(-30, 1, (e,), device='cuda:0')
    for ex in (None, extra):
        want = rel_geometry_torch(pos_s[:, :2], pos_d, head_s, head_d, ei, ex)   # the kernel takes the row stride
        got = ops.rel_geometry(pos_s, pos_d, head_s, head_d, ei, ex)
        assert got.shape == want.shape
        assert torch.equal(got[:50, 1], torch.zeros(50, device='cuda:0'))       # angle of a zero vector is +0
        assert float((got - want).abs().max()) < 2e-6
        assert (got == want).float().mean() > 0.99
    empty = ops.rel_geometry(pos_s, pos_d, head_s, head_d, ei[:, :0])
    assert empty.shape == (0, 3)


@pytest.mark.gpu
def test_rel_geometry_kernel_vs_reference_golden():
    """qcn_rel_geometry_f32 against rows the REFERENCE's own utils/geometry.py produced (pinned: tests/golden/
    ref_geometry_edges.npz, written by oracle/make_golden.py::main_geometry_edges in the reference's call pattern,
    modules/qcnet_agent_encoder.py:161-166).  The kernel evaluates cos / sin / atan2 / sqrt on the GPU, the golden rows
    come from the host's libm: agreement to 2e-6 absolute (angles, radians) and 1e-6 relative (distances); the
    signed-zero rows (zero displacement: angle +0, not pi) and the wrap boundary must agree exactly."""
    import os
    import numpy as np
    from tempenslearnagg_qcnet_amd import ops
    E = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_geometry_edges.npz'))
    pos_s, pos_d, head_s, head_d, ei, want = (torch.from_numpy(E[k]).to('cuda:0') for k in
                                               ('pos_s', 'pos_d', 'head_s', 'head_d', 'edge_index', 'rows'))
    got = ops.rel_geometry(pos_s.contiguous(), pos_d.contiguous(), head_s, head_d, ei, None)
    assert got.shape == want.shape
    assert torch.equal(got[:40, :2], want[:40, :2])                           # zero displacement: distance 0, angle +0
    assert float(((got[:, 0] - want[:, 0]).abs() / want[:, 0].clamp_min(1e-3)).max()) < 1e-6
    assert float((got[:, 1:] - want[:, 1:]).abs().max()) < 2e-6
    assert torch.equal(got[40:70, 2] == 0, want[40:70, 2] == 0)               # equal headings: exactly zero where the reference's is


@pytest.mark.parametrize('b,n,m,band,p_src,p_dst', [
    (7, 59, 59, (1, 10), 0.8, 0.8),          # temporal graph (modules/qcnet_agent_encoder.py:129-132)
    (50, 50, 6, None, 0.6, 0.9),             # time -> mode (bipartite, modules/qcnet_decoder.py:160-169)
    (33, 6, 6, None, 0.7, 0.7),              # mode -> mode, self loops kept (:213)
    (4, 300, 300, (1, 300), 0.5, 0.5),       # time_span >= n: every later step
    (3, 5, 9, (-2, 3), 1.0, 1.0),            # a band reaching both sides, bipartite
    (2, 8, 8, None, 0.0, 1.0),               # nothing valid
])
def test_mask_pairs_bit_exact_vs_oracle_dense_to_sparse(b, n, m, band, p_src, p_dst):
    g = torch.Generator().manual_seed(b * 1000 + n)
    ms = torch.rand(b, n, generator=g) < p_src
    md = ms.clone() if (n == m and p_src == p_dst) else (torch.rand(b, m, generator=g) < p_dst)
    adj = ms.unsqueeze(2) & md.unsqueeze(1)
    want = oracle.dense_to_sparse(adj) if n == m else oracle.bipartite_dense_to_sparse(adj)
    if n == m:
        assert torch.equal(want, oracle.bipartite_dense_to_sparse(adj))
    if band is not None:      # the reference's two filter passes (modules/qcnet_agent_encoder.py:131-132), generalised
        i, j = want[0] % n, want[1] % m
        want = want[:, (j - i >= band[0]) & (j - i <= band[1])]
    lo, hi = band if band is not None else (None, None)
    got, offset = ops.mask_pairs(ms.to(DEV), md.to(DEV), lo, hi)
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    assert int(offset[-1]) == want.size(1) and offset.numel() == b * n + 1
    assert torch.equal(G.mask_outer_edges(ms.to(DEV), md.to(DEV), lo, hi).cpu(), want)


def test_mask_pairs_empty_and_bad_inputs():
    e, _ = ops.mask_pairs(torch.zeros(0, 5, dtype=torch.bool, device=DEV), torch.zeros(0, 3, dtype=torch.bool, device=DEV))
    assert e.shape == (2, 0)
    with pytest.raises(ValueError):
        ops.mask_pairs(torch.zeros(2, 5, dtype=torch.bool, device=DEV), torch.zeros(3, 5, dtype=torch.bool, device=DEV))
    with pytest.raises(pkg._lib.QcnError):
        ops.mask_pairs(torch.ones(2, 5, dtype=torch.bool, device=DEV), torch.ones(2, 5, dtype=torch.bool, device=DEV), 3, 1)


@pytest.mark.parametrize('n_nodes,E,with_attr', [(30, 400, True), (4800, 200000, True), (150, 5000, False), (1, 10, True),
                                               (1 << 20, 3000, True)])
def test_coalesce_max_kernel_bit_exact_vs_oracle(n_nodes, E, with_attr):
    g = torch.Generator().manual_seed(E)
    ei = torch.randint(0, n_nodes, (2, E), generator=g)
    ei[:, E // 2:] = ei[:, :E - E // 2]                       # every edge of the first half again (different types)
    attr = torch.randint(0, 5, (E,), generator=g).to(torch.uint8)
    wi, wa = oracle.coalesce_max(ei, attr)
    gi, ga = ops.coalesce_max(ei.to(DEV), attr.to(DEV) if with_attr else None, n_nodes)
    assert torch.equal(gi.cpu(), wi)
    if with_attr:
        assert ga.dtype == torch.uint8 and torch.equal(ga.cpu(), wa)
    else:
        assert ga is None
    # the public utility takes the same route and keeps PyG's return conventions
    assert torch.equal(G.coalesce(ei.to(DEV), num_nodes=n_nodes).cpu(), wi)
    ci, ca = G.coalesce(ei.to(DEV), attr.to(DEV), reduce='max')
    assert torch.equal(ci.cpu(), wi) and torch.equal(ca.cpu(), wa)
    with pytest.raises(IndexError):
        ops.coalesce_max(ei.to(DEV) + n_nodes, None, n_nodes)
    empty, _ = ops.coalesce_max(torch.empty(2, 0, dtype=torch.long, device=DEV), None, 5)
    assert empty.shape == (2, 0)
