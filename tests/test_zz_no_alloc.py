"""GPU: qcn_csr_build_static — the allocation-free variant of qcn_csr_build (sort workspace in a device array inside the
library) — gives bit-identical CSR arrays to qcn_csr_build, on two calls in a row (the second smaller, so stale scratch
contents of the first would show), rejects sizes beyond its capacity, and performs no device allocation."""
import ctypes

import pytest
import torch

from tempenslearnagg_qcnet_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _edges(num_edges, n_src, n_dst, seed):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n_src, (num_edges,), generator=g)
    dst = torch.randint(0, n_dst, (num_edges,), generator=g)
    return torch.stack([src, dst]).to(DEV)


def _static(ei, n_dst):
    e = ei.size(1)
    rowptr = torch.full((n_dst + 1,), -7, dtype=torch.int32, device=DEV)
    col = torch.full((e,), -7, dtype=torch.int32, device=DEV)
    eid = torch.full((e,), -7, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(DEV)['allocation.all.allocated'], torch.cuda.mem_get_info(DEV)[0]
    ops.csr_build_static(ei, n_dst, rowptr, col, eid)
    torch.cuda.synchronize()
    after = torch.cuda.memory_stats(DEV)['allocation.all.allocated'], torch.cuda.mem_get_info(DEV)[0]
    return (rowptr, col, eid), before, after


def test_csr_build_static_matches_csr_build_twice_in_a_row():
    torch.cuda.set_device(0)
    cases = [(200_000, 5_000, 3_000, 1), (777, 40, 2_000, 2), (1, 3, 3, 3), (0, 5, 9, 4)]   # second call smaller than the first
    warm = _edges(16, 4, 4, 0)
    _static(warm, 4)                                 # first call on the device: looks up the scratch address
    for num_edges, n_src, n_dst, seed in cases:
        ei = _edges(num_edges, n_src, n_dst, seed)
        want = ops.csr_build(ei, n_dst)
        got, before, after = _static(ei, n_dst)
        for g, w, name in zip(got, want, ('rowptr', 'col', 'eid')):
            assert torch.equal(g, w), (name, num_edges)
        assert after[0] == before[0], 'torch allocation inside the allocation-free entry point'
        assert after[1] >= before[1], 'free device memory shrank across the call: something allocated'
        if num_edges:
            # CSR invariants, independent of the other entry point: stable, destination-sorted
            dst_sorted = ei[1][got[2].long()]
            assert bool((dst_sorted[1:] >= dst_sorted[:-1]).all())
            same = dst_sorted[1:] == dst_sorted[:-1]
            assert bool((got[2][1:][same] > got[2][:-1][same]).all())
            assert torch.equal(got[1].long(), ei[0][got[2].long()])


def test_csr_build_static_at_capacity_and_beyond():
    torch.cuda.set_device(0)
    cap = ops.csr_static_max_edges()
    assert cap == 1 << 20
    ei = _edges(cap, 1 << 16, 1 << 18, 5)
    want = ops.csr_build(ei, 1 << 18)
    got, _, _ = _static(ei, 1 << 18)
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    big = _edges(cap + 1, 10, 10, 6)
    out = [torch.zeros(11, dtype=torch.int32, device=DEV), torch.zeros(cap + 1, dtype=torch.int32, device=DEV),
           torch.zeros(cap + 1, dtype=torch.int32, device=DEV)]
    with pytest.raises(_lib.QcnError, match='capacity'):
        ops.csr_build_static(big, 10, *out)
    assert int(out[0].abs().sum()) == 0                # nothing was launched
    assert _lib.lib().qcn_csr_build_static(None, None, ctypes.c_int64(-1), 4, None, None, None, None) == -1
