"""Graph construction for the QCNet path: MI355X-native counterparts of the torch_cluster / torch_geometric
utilities the reference modules call, with the same signatures and result order.

* ``radius`` / ``radius_graph``  — torch_cluster 1.6.1 signatures (modules/qcnet_agent_encoder.py:158,168);
  canonical result order of SURVEY.md §8 G1/G2 (y ascending, x ascending inside a y, lowest-index survivors
  under the neighbour cap).
* ``dense_to_sparse`` / ``bipartite_dense_to_sparse`` / ``subgraph`` / ``coalesce`` / ``merge_edges`` —
  torch_geometric 2.3.0 semantics and utils/graph.py:45-54,88-94.
* ``CSRGraph`` — the destination-sorted CSR every AttentionLayer call consumes.  The reference rebuilds its
  gather/scatter indexing from COO inside every propagate call; here a relation's CSR is built once and
  shared by all layers that use the relation.
"""
from typing import List, Optional, Sequence, Tuple

import torch

from .. import ops


class CSRGraph:
    """A bipartite relation src → dst in two forms: the reference-order COO ``edge_index`` ([2, E] int64)
    and a destination-sorted CSR (``rowptr`` int32 [n_rows + 1], ``col`` int32 [E_base] source per slot,
    ``eid`` int32 [E_base] original edge id per slot, or None when the COO order already is the CSR order).

    Mode replication: the decoder attends with K mode queries per agent that share one edge list and one
    positional embedding (modules/qcnet_decoder.py:170,190-192,208-211).  Such a relation is stored once
    (``n_rows`` CSR rows, ``E_base`` edges) with ``rep = K`` and a ``layout`` telling where mode ``m`` of row
    ``i`` lives in the destination tensor: ``LAYOUT_MODE_MAJOR`` → ``m * n_rows + i``, ``LAYOUT_AGENT_MAJOR`` →
    ``i * rep + m``.  ``edge_index`` still holds the reference's replicated COO list (E = E_base * rep)."""

    __slots__ = ('edge_index', 'n_src', 'n_dst', 'rowptr', 'col', 'eid', 'rep', 'layout', 'col_shift', 'num_feature_rows', 'row_index', 'src_offset')

    def __init__(self, edge_index, n_src, n_dst, rowptr, col, eid, rep: int = 1, layout: int = ops.LAYOUT_PLAIN,
                 col_shift: int = 0, num_feature_rows: Optional[int] = None):
        self.edge_index = edge_index
        self.n_src = int(n_src)
        self.n_dst = int(n_dst)
        self.rowptr = rowptr
        self.col = col
        self.eid = eid
        self.rep = int(rep)
        self.layout = int(layout)
        # windows of a larger (absolute-time) relation: rowptr is a view with absolute offsets into col / eid, source
        # ids are re-based by col_shift and the positional-feature table has num_feature_rows rows
        self.col_shift = int(col_shift)
        self.num_feature_rows = int(num_feature_rows) if num_feature_rows is not None else int(col.numel())
        # per-destination indirections (with_rows): CSR row and source re-basing of every destination, int32 [n_dst]
        self.row_index = None
        self.src_offset = None

    @property
    def num_base_edges(self) -> int:
        return int(self.col.numel())

    @property
    def num_edges(self) -> int:
        return int(self.col.numel()) * self.rep

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, n_src: int, n_dst: int, validate: bool = True) -> 'CSRGraph':
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError('edge_index must be [2, E]')
        edge_index = edge_index.contiguous()
        if validate and edge_index.size(1) > 0:
            # One host sync; keeps a malformed edge list from becoming an out-of-bounds gather on the GPU.
            lo = int(edge_index.min())
            hi_src, hi_dst = int(edge_index[0].max()), int(edge_index[1].max())
            if lo < 0 or hi_src >= n_src or hi_dst >= n_dst:
                raise IndexError('edge_index out of range: min %d, max src %d (n_src %d), max dst %d (n_dst %d)'
                                 % (lo, hi_src, n_src, hi_dst, n_dst))
        rowptr, col, eid = ops.csr_build(edge_index, n_dst)
        return cls(edge_index, n_src, n_dst, rowptr, col, eid)

    @classmethod
    def from_sorted(cls, edge_index: torch.Tensor, offset: torch.Tensor, n_src: int, n_dst: int) -> 'CSRGraph':
        """``edge_index`` already grouped by ascending destination with row pointer ``offset`` (int64)."""
        return cls(edge_index, n_src, n_dst, offset.to(torch.int32), edge_index[0].to(torch.int32), None)

    def fold_sources(self, period: int) -> 'CSRGraph':
        """Source rows that are ``period``-periodic copies (the reference tiles ``x_pl`` T times,
        modules/qcnet_map_encoder.py:170, and K times, modules/qcnet_decoder.py:150-151) are gathered from one
        copy: source index modulo ``period``.  ``edge_index`` keeps the reference numbering."""
        return CSRGraph(self.edge_index, period, self.n_dst, self.rowptr, torch.remainder(self.col, period), self.eid,
                        self.rep, self.layout, self.col_shift, self.num_feature_rows)

    def window(self, row_begin: int, row_end: int, n_src: int, col_shift: int = 0) -> 'CSRGraph':
        """Rows [row_begin, row_end) as a graph of their own without copying anything: the row pointer is a view
        (offsets stay absolute), sources are re-based by ``col_shift``.  ``edge_index`` is not carried over."""
        assert self.rep == 1
        return CSRGraph(None, n_src, row_end - row_begin, self.rowptr[row_begin:row_end + 1], self.col, self.eid,
                        col_shift=col_shift, num_feature_rows=self.num_feature_rows)

    def remap_sources(self, col: torch.Tensor, n_src: int) -> 'CSRGraph':
        """The same edges reading from a re-indexed source tensor (``col``: new source row per CSR slot)."""
        g = CSRGraph(self.edge_index, n_src, self.n_dst, self.rowptr, col.to(torch.int32).contiguous(), self.eid,
                     self.rep, self.layout, self.col_shift, self.num_feature_rows)
        g.row_index, g.src_offset = self.row_index, self.src_offset
        return g

    def with_rows(self, row_index: torch.Tensor, src_offset: Optional[torch.Tensor], n_src: int) -> 'CSRGraph':
        """A new destination set over this CSR without copying it: destination ``i`` uses CSR row ``row_index[i]`` and
        reads source rows ``col + src_offset[i]`` (None: unchanged) of an ``n_src``-row source tensor."""
        assert self.rep == 1 and self.row_index is None and self.col_shift == 0
        g = CSRGraph(None, n_src, row_index.numel(), self.rowptr, self.col, self.eid,
                     num_feature_rows=self.num_feature_rows)
        g.row_index = row_index.to(torch.int32).contiguous()
        g.src_offset = None if src_offset is None else src_offset.to(torch.int32).contiguous()
        return g

    def frames(self, n_frames: int, rows_per_frame: int, row_shift: int, n_src: int, col_shift: int = 0,
               src_stride: int = 0) -> 'CSRGraph':
        """All sliding frames of an absolute-time relation as ONE graph without copying: frame f owns destinations
        [f * rows_per_frame, (f + 1) * rows_per_frame), which are CSR rows [f * row_shift, f * row_shift + rows_per_frame);
        its source ids are col - f * col_shift + f * src_stride."""
        dev = self.rowptr.device
        f = torch.arange(n_frames, device=dev, dtype=torch.int64).unsqueeze(1)
        r = torch.arange(rows_per_frame, device=dev, dtype=torch.int64).unsqueeze(0)
        row_index = (r + f * row_shift).reshape(-1)
        src_offset = (f * (src_stride - col_shift)).expand(n_frames, rows_per_frame).reshape(-1)
        return self.with_rows(row_index, src_offset if (col_shift or src_stride) else None, n_src)

    def validate(self) -> None:
        """Host check (several syncs) that every destination's CSR row and every source row it reads are in range."""
        n_rows = self.rowptr.numel() - 1
        rows = self.row_index.long() if self.row_index is not None else torch.arange(n_rows, device=self.rowptr.device)
        if rows.numel() and (int(rows.min()) < 0 or int(rows.max()) >= n_rows):
            raise IndexError('row_index out of range')
        beg, end = self.rowptr.long()[rows], self.rowptr.long()[rows + 1]
        if rows.numel() and (int(beg.min()) < 0 or int(end.max()) > self.col.numel() or bool((end < beg).any())):
            raise IndexError('rowptr out of range')
        if self.col.numel() == 0 or rows.numel() == 0:
            return
        colf = self.col.long()
        # per-row min / max through segment reductions on the (small) set of distinct CSR rows
        lens = (self.rowptr.long()[1:] - self.rowptr.long()[:-1])
        seg = torch.repeat_interleave(torch.arange(n_rows, device=colf.device), lens)
        rmin = torch.full((n_rows,), 1 << 40, device=colf.device, dtype=torch.int64).scatter_reduce(0, seg, colf, 'amin')
        rmax = torch.full((n_rows,), -(1 << 40), device=colf.device, dtype=torch.int64).scatter_reduce(0, seg, colf, 'amax')
        off = (self.src_offset.long() if self.src_offset is not None else 0) - self.col_shift
        has = end > beg
        lo, hi = (rmin[rows] + off)[has], (rmax[rows] + off)[has]
        if lo.numel() and (int(lo.min()) < 0 or int(hi.max()) >= self.n_src):
            raise IndexError('source rows out of range: [%d, %d] for n_src %d' % (int(lo.min()), int(hi.max()), self.n_src))
        if self.eid is not None and self.eid.numel() and (int(self.eid.min()) < 0 or int(self.eid.max()) >= self.num_feature_rows):
            raise IndexError('eid out of range')

    def replicate(self, rep: int, layout: int, edge_index: Optional[torch.Tensor] = None) -> 'CSRGraph':
        """The same edge list for ``rep`` mode copies of every destination row (see class docstring).
        ``edge_index``: the reference's replicated COO list, kept for inspection / parity checks only."""
        assert self.rep == 1 and layout in (ops.LAYOUT_MODE_MAJOR, ops.LAYOUT_AGENT_MAJOR)
        return CSRGraph(edge_index if edge_index is not None else self.edge_index, self.n_src, self.n_dst * rep,
                        self.rowptr, self.col, self.eid, rep, layout)


def _ptr_from_batch(batch: Optional[torch.Tensor], n: int, device) -> torch.Tensor:
    if batch is None:
        return torch.tensor([0, n], dtype=torch.int64, device=device)
    batch_size = int(batch.max()) + 1 if batch.numel() > 0 else 1
    return torch.bucketize(torch.arange(batch_size + 1, device=device, dtype=torch.int64), batch)


def _align_ptrs(ptr_x: torch.Tensor, ptr_y: torch.Tensor, n_x: int, n_y: int) -> Tuple[torch.Tensor, torch.Tensor]:
    # batch_size = max over both batch vectors (torch_cluster/radius.py); pad the shorter ptr with its total.
    if ptr_x.numel() < ptr_y.numel():
        ptr_x = torch.cat([ptr_x, ptr_x.new_full((ptr_y.numel() - ptr_x.numel(),), n_x)])
    elif ptr_y.numel() < ptr_x.numel():
        ptr_y = torch.cat([ptr_y, ptr_y.new_full((ptr_x.numel() - ptr_y.numel(),), n_y)])
    return ptr_x, ptr_y


def radius(x: torch.Tensor, y: torch.Tensor, r: float, batch_x: Optional[torch.Tensor] = None,
           batch_y: Optional[torch.Tensor] = None, max_num_neighbors: int = 32, num_workers: int = 1) -> torch.Tensor:
    """Drop-in for ``torch_cluster.radius``: [2, E] int64, row0 = y index, row1 = x index."""
    if x.numel() == 0 or y.numel() == 0:
        return torch.empty(2, 0, dtype=torch.long, device=x.device)
    x = x.reshape(-1, 1) if x.dim() == 1 else x
    y = y.reshape(-1, 1) if y.dim() == 1 else y
    x, y = x.contiguous().float(), y.contiguous().float()
    if (batch_x is None) != (batch_y is None):
        raise AssertionError('batch_x and batch_y must be given together')
    ptr_x = _ptr_from_batch(batch_x, x.size(0), x.device)
    ptr_y = _ptr_from_batch(batch_y, y.size(0), x.device)
    ptr_x, ptr_y = _align_ptrs(ptr_x, ptr_y, x.size(0), y.size(0))
    edge_index, _ = ops.radius_pairs(x, y, r, ptr_x, ptr_y, max_num_neighbors)
    return edge_index


def radius_graph(x: torch.Tensor, r: float, batch: Optional[torch.Tensor] = None, loop: bool = False,
                 max_num_neighbors: int = 32, flow: str = 'source_to_target', num_workers: int = 1) -> torch.Tensor:
    """Drop-in for ``torch_cluster.radius_graph``: row0 = neighbour (source), row1 = query (target) for
    ``source_to_target``; the neighbour cap is ``max_num_neighbors + 1`` hits including the self pair."""
    assert flow in ('source_to_target', 'target_to_source')
    if x.numel() == 0:
        return torch.empty(2, 0, dtype=torch.long, device=x.device)
    x = x.reshape(-1, 1) if x.dim() == 1 else x
    x = x.contiguous().float()
    ptr = _ptr_from_batch(batch, x.size(0), x.device)
    cap = max_num_neighbors if loop else max_num_neighbors + 1
    edge_index, _ = ops.radius_pairs(x, x, r, ptr, ptr, cap, drop_self=not loop,
                                     y_first=(flow != 'source_to_target'))
    return edge_index


def dense_to_sparse(adj: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``torch_geometric.utils.dense_to_sparse`` for a [B, n, n] (or [n, n]) adjacency: row-major nonzero
    order, node ids offset by ``b * n``.  Returns ``(edge_index, edge_attr)`` like PyG."""
    if adj.dim() == 2:
        idx = adj.nonzero(as_tuple=True)
        return torch.stack(idx, dim=0), adj[idx]
    b, i, j = adj.nonzero(as_tuple=True)
    return torch.stack([b * adj.size(1) + i, b * adj.size(2) + j], dim=0), adj[b, i, j]


def mask_outer_edges(mask_src: torch.Tensor, mask_dst: torch.Tensor, delta_lo: Optional[int] = None,
                     delta_hi: Optional[int] = None) -> torch.Tensor:
    """``bipartite_dense_to_sparse(mask_src.unsqueeze(2) & mask_dst.unsqueeze(1))`` (= ``dense_to_sparse`` for square
    masks), optionally restricted to ``delta_lo <= j - i <= delta_hi``, from the HIP enumeration kernel: the dense
    [B, n, m] adjacency of the reference (modules/qcnet_agent_encoder.py:129-132, modules/qcnet_decoder.py:160-169,213)
    is never built.  Same edge order as the reference."""
    return ops.mask_pairs(mask_src, mask_dst, delta_lo, delta_hi)[0]


def bipartite_dense_to_sparse(adj: torch.Tensor) -> torch.Tensor:
    """utils/graph.py:88-94 — [B, n, m] mask → src = b*n + i, dst = b*m + j."""
    idx = adj.nonzero(as_tuple=True)
    if len(idx) == 3:
        return torch.stack([idx[0] * adj.size(1) + idx[1], idx[0] * adj.size(2) + idx[2]], dim=0)
    return torch.stack(idx, dim=0)


def subgraph(subset: torch.Tensor, edge_index: torch.Tensor,
             edge_attr: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """``torch_geometric.utils.subgraph`` with a bool node mask and ``relabel_nodes=False``."""
    if subset.dtype != torch.bool:
        mask = torch.zeros(int(edge_index.max()) + 1 if edge_index.numel() else 0, dtype=torch.bool,
                           device=edge_index.device)
        mask[subset] = True
        subset = mask
    keep = subset[edge_index[0]] & subset[edge_index[1]]
    return edge_index[:, keep], (edge_attr[keep] if edge_attr is not None else None)


def coalesce(edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor] = None, num_nodes: Optional[int] = None,
             reduce: str = 'add') -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """``torch_geometric.utils.coalesce``: sort by (row, col), merge duplicates with ``reduce``."""
    if edge_index.size(1) == 0:
        return (edge_index, edge_attr) if edge_attr is not None else edge_index
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    if edge_index.is_cuda and edge_index.dtype == torch.int64 and n <= (1 << 27) and (
            edge_attr is None or (reduce == 'max' and edge_attr.dtype == torch.uint8 and edge_attr.dim() == 1)):
        # the reference's case (utils/graph.py:54: reduce='max' over uint8 edge types): HIP sort + merge
        out_index, out_attr = ops.coalesce_max(edge_index, edge_attr, n)
        return out_index if edge_attr is None else (out_index, out_attr)
    key = edge_index[0] * n + edge_index[1]
    uniq, inverse = torch.unique(key, sorted=True, return_inverse=True)
    out_index = torch.stack([torch.div(uniq, n, rounding_mode='floor'), uniq % n], dim=0)
    if edge_attr is None:
        return out_index
    torch_reduce = {'add': 'sum', 'sum': 'sum', 'max': 'amax', 'min': 'amin', 'mean': 'mean'}[reduce]
    work = edge_attr.long() if not edge_attr.is_floating_point() else edge_attr
    out_attr = work.new_zeros((uniq.numel(),) + tuple(work.shape[1:]))
    idx = inverse.view(-1, *([1] * (work.dim() - 1))).expand_as(work)
    out_attr.scatter_reduce_(0, idx, work, reduce=torch_reduce, include_self=False)
    return out_index, out_attr.to(edge_attr.dtype)


def merge_edges(edge_indices: List[torch.Tensor], edge_attrs: Optional[List[torch.Tensor]] = None,
                reduce: str = 'add') -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """utils/graph.py:45-54."""
    edge_index = torch.cat(edge_indices, dim=1)
    edge_attr = torch.cat(edge_attrs, dim=0) if edge_attrs is not None else None
    if edge_attr is None:
        return coalesce(edge_index, None, reduce=reduce), None
    return coalesce(edge_index, edge_attr, reduce=reduce)
