"""AttentionLayer — drop-in for the reference's PyG ``MessagePassing`` block (layers/attention_layer.py:24-122).

Same constructor arguments, same ``forward(x, r, edge_index)`` contract and the same ``state_dict`` keys
(including the aliased ``attn_prenorm_x_dst`` of non-bipartite layers, reference :60-61), but no PyG: the
gather → segment softmax → scatter-add of ``propagate`` is one HIP kernel over a destination-sorted CSR
(``qcn_attn_fwd_f32``).  Inference only — dropout sites (:49,53,99) are identities in eval mode and the layer
refuses to run in training mode with ``dropout > 0``.
"""
import os
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops
from ..utils.graph import CSRGraph
from ..utils.weight_init import weight_init


# QCN_ATTN_MODES=0 routes mode-replicated relations through qcn_attn_fused_f32 instead of qcn_attn_modes_f32 (A/B runs)
USE_ATTN_MODES = os.environ.get('QCN_ATTN_MODES', '1') != '0'


# keep the per-edge projections of a mode-replicated relation across repeated applications of a layer to the same sources
# (qcn_attn_modes_store_* / qcn_attn_modes_cached_f32; QCN_EDGE_KV_CACHE=0: recompute them every time)
# Byte budget of ONE (layer, relation)'s pair of tables (1 KiB per base edge; six layer-relations are alive together during
# a proposal phase, TemporalAggregate's overlapped half adds its own): beyond it — or when the allocator reports less than
# twice that free — the layer recomputes the projections (no silent exhaustion of HBM at cfg4-like sizes).
EDGE_TABLE_BUDGET_BYTES = int(os.environ.get('QCN_EDGE_KV_BUDGET_MB', '8192')) << 20


def _tables_fit(graph, device) -> bool:
    need = AttentionLayer.edge_table_bytes(graph)
    if need > EDGE_TABLE_BUDGET_BYTES or ops.attn_modes_table_slots(graph.num_base_edges, graph.rowptr.numel() - 1) >= (1 << 23):
        return False
    if need > (256 << 20):          # only large tables are worth a driver query (cfg3: 2.1 GB for pl2m, 0.5 / 0.7 GB others)
        free, _ = torch.cuda.mem_get_info(device)
        reserved_free = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        return need * 2 < free + reserved_free
    return True


CACHE_EDGE_PROJECTIONS = os.environ.get('QCN_EDGE_KV_CACHE', '1') != '0'

# node side of a layer as ONE launch behind the propagate (qcn_attn_post_chain_f32: to_s computed in the kernel, the next
# layer's destination-side projections applied to the output row); QCN_ATTN_CHAIN=0: qcn_ln_linear_f32 writes s, the next
# layer projects its own destination rows (A/B runs)
CHAIN_NODE_SIDE = os.environ.get('QCN_ATTN_CHAIN', '1') != '0'


class NormalizedEdgeFeatures:
    """Positional embeddings of a relation after the layer-independent half of ``attn_prenorm_r``:
    ``rhat = (r - mean) / sqrt(var + eps)`` per edge, [E_base, 128].  Every AttentionLayer that consumes the
    relation applies only its own affine (folded into its projection weights), so the normalisation is done
    once per relation instead of once per layer (reference: layers/attention_layer.py:82 in every call).
    Held in whichever format it was produced in — fp32 rows, split-f16 rows (``normalized_output='split'``) or bf16 rows
    (``'bf16'``) — and converted on first use to what a kernel asks for."""

    __slots__ = ('rhat', '_split', '_bf16')

    def __init__(self, rhat: torch.Tensor) -> None:
        self.rhat = self._split = self._bf16 = None
        if rhat.dtype == torch.float16:      # split operand format (FourierEmbedding normalized_output='split')
            self._split = rhat
        elif rhat.dtype == torch.bfloat16:   # bf16 operand format (normalized_output='bf16')
            self._bf16 = rhat
        else:
            self.rhat = rhat

    @classmethod
    def deferred(cls, make, inputs=(), rep: Optional[int] = None) -> 'NormalizedEdgeFeatures':
        """The features ``make()`` returns (one large Fourier-embedding launch over tensors ``inputs``), produced where
        utils/prep.py::defer puts heavy launches: at once outside a preparation section, on the main stream right after
        the section inside one.  ``rep``: convert for a relation with that many queries per row (prepare_for)."""
        from ..utils import prep
        self = cls.__new__(cls)
        self.rhat = self._split = self._bf16 = None

        def fill():
            cls.__init__(self, make())
            if rep is not None:
                self.prepare_for(rep)
        prep.defer(fill, inputs)
        return self

    @property
    def num_rows(self) -> int:
        return int(next(t for t in (self.rhat, self._split, self._bf16) if t is not None).size(0))

    @property
    def split(self) -> torch.Tensor:
        """The same rows as split f16 ([E_base, 256], qcn_split_rows_f16): the operand format of qcn_attn_modes_f32,
        converted once per relation on first use."""
        if self._split is None:
            self._split = ops.split_rows(self.dense)
        return self._split

    @property
    def bf16(self) -> torch.Tensor:
        """The same rows rounded to bf16 ([E_base, 128], round to nearest even): the operand of qcn_attn_modes_bf16."""
        if self._bf16 is None:
            self._bf16 = self.dense.bfloat16()
        return self._bf16

    @property
    def dense(self) -> torch.Tensor:
        """fp32 rows [E_base, 128] (the operand of qcn_attn_fused_f32)."""
        if self.rhat is None:
            if self._split is not None:
                sp = self._split.float()
                self.rhat = ((sp[:, :ops.C] + sp[:, ops.C:]) * 2.0 ** -ops.RHAT_SPLIT_SCALE_LOG2).contiguous()
            else:
                self.rhat = self._bf16.float()
        return self.rhat

    def prepare_for(self, rep: int, mode: Optional[int] = None) -> None:
        """Convert ahead of the layers (on the preparation stream) what a relation with ``rep`` queries per row needs."""
        if USE_ATTN_MODES and 1 < rep <= 6:
            self.bf16 if ops._mode(mode) == ops.MATMUL_BF16 else self.split  # noqa: B018  (cached properties)

    @classmethod
    def from_raw(cls, r: torch.Tensor, eps: float = 1e-5) -> 'NormalizedEdgeFeatures':
        return cls(ops.row_normalize(r.contiguous(), eps))


class AttentionLayer(nn.Module):

    def __init__(self,
                 hidden_dim: int,
                 num_heads: int,
                 head_dim: int,
                 dropout: float,
                 bipartite: bool,
                 has_pos_emb: bool,
                 **kwargs) -> None:
        super().__init__()
        if hidden_dim != ops.C or num_heads * head_dim != ops.C or head_dim != 16:
            raise NotImplementedError('the gfx950 kernels are built for hidden_dim = 128 = 8 heads x 16 '
                                      '(got hidden_dim=%d, %d x %d)' % (hidden_dim, num_heads, head_dim))
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.has_pos_emb = has_pos_emb
        self.bipartite = bipartite
        self.dropout = dropout
        self.scale = head_dim ** -0.5
        inner = head_dim * num_heads

        self.to_q = nn.Linear(hidden_dim, inner)
        self.to_k = nn.Linear(hidden_dim, inner, bias=False)
        self.to_v = nn.Linear(hidden_dim, inner)
        if has_pos_emb:
            self.to_k_r = nn.Linear(hidden_dim, inner, bias=False)
            self.to_v_r = nn.Linear(hidden_dim, inner)
        self.to_s = nn.Linear(hidden_dim, inner)
        self.to_g = nn.Linear(inner + hidden_dim, inner)
        self.to_out = nn.Linear(inner, hidden_dim)
        self.attn_drop = nn.Dropout(dropout)
        self.ff_mlp = nn.Sequential(
            nn.Linear(hidden_dim, hidden_dim * 4),
            nn.ReLU(inplace=True),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim * 4, hidden_dim),
        )
        self.attn_prenorm_x_src = nn.LayerNorm(hidden_dim)
        self.attn_prenorm_x_dst = nn.LayerNorm(hidden_dim) if bipartite else self.attn_prenorm_x_src
        if has_pos_emb:
            self.attn_prenorm_r = nn.LayerNorm(hidden_dim)
        self.attn_postnorm = nn.LayerNorm(hidden_dim)
        self.ff_prenorm = nn.LayerNorm(hidden_dim)
        self.ff_postnorm = nn.LayerNorm(hidden_dim)
        self.apply(weight_init)
        self._pack_key = None
        self._packs = None

    def _kernel_packs(self):
        """All parameters in the layouts the kernels consume (include/qcnet_hip.h), cached per parameter version:
        ``pre_*``: chain-packed projection stacks for qcn_ln_linear_f32; ``post``: qcn_attn_post_f32 pack;
        ``pos``: (wk, wv, bv) for qcn_attn_fused_f32 with the attn_prenorm_r affine folded in
        (wk = to_k_r.weight * gamma, wv = to_v_r.weight * gamma, bv = to_v_r.weight @ beta + to_v_r.bias)."""
        mode = ops.matmul_mode()
        key = (mode,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if key == self._pack_key:
            return self._packs
        C = ops.C
        with torch.no_grad():
            zero = torch.zeros(C, device=self.to_q.weight.device)
            stack = lambda *lins: ops.pack_chain_matrices([lin.weight for lin in lins], mode=mode)
            packs = {'mode': mode}
            packs['pre_src'] = (stack(self.to_k, self.to_v), torch.stack([zero, self.to_v.bias]).contiguous())
            packs['pre_dst'] = (stack(self.to_q, self.to_s),
                                torch.stack([self.to_q.bias, self.to_s.bias]).contiguous())
            packs['pre_all'] = (stack(self.to_q, self.to_k, self.to_v, self.to_s),
                                torch.stack([self.to_q.bias, zero, self.to_v.bias, self.to_s.bias]).contiguous())
            # chained node side (qcn_attn_post_chain_f32 computes s): what is left for qcn_ln_linear_f32 / the previous layer
            packs['pre_q'] = (stack(self.to_q), self.to_q.bias.reshape(1, C).contiguous())
            # static operand scale of the LayerNorm output in front of the destination-side projections (the previous layer's
            # node-side kernel applies them in a chained stack)
            ln_d = self.attn_prenorm_x_dst if self.bipartite else self.attn_prenorm_x_src
            packs['dst_operand_scale'] = ops.static_operand_scale(ops.layer_norm_bound(ln_d.weight, ln_d.bias))
            packs['pre_qkv'] = (stack(self.to_q, self.to_k, self.to_v),
                                torch.stack([self.to_q.bias, zero, self.to_v.bias]).contiguous())
            w1, w2 = self.ff_mlp[0], self.ff_mlp[3]
            ffn = []
            for hb in range(4):
                ffn += [w1.weight[hb * C:(hb + 1) * C, :], w2.weight[:, hb * C:(hb + 1) * C]]
            vecs = [self.to_g.bias, self.to_out.bias, self.attn_postnorm.weight, self.attn_postnorm.bias,
                    self.ff_prenorm.weight, self.ff_prenorm.bias, w1.bias, w2.bias, self.ff_postnorm.weight,
                    self.ff_postnorm.bias, self.attn_prenorm_x_dst.weight, self.attn_prenorm_x_dst.bias]
            mats, scales = ops.pack_chain_set([self.to_g.weight[:, :C], self.to_g.weight[:, C:], self.to_out.weight] + ffn, mode=mode)
            packs['post'] = torch.cat([mats.reshape(-1)] + [v.reshape(-1).float() for v in vecs]
                                      + [ops.scale_table(scales)]).contiguous()
            ln = lambda m: (m.weight, m.bias)
            packs['post_chain'] = ops.pack_attn_post_chain(
                self.to_g.weight, self.to_g.bias, self.to_s.weight, self.to_s.bias, self.to_out.weight, self.to_out.bias,
                w1.weight, w1.bias, w2.weight, w2.bias, ln(self.attn_postnorm), ln(self.ff_prenorm), ln(self.ff_postnorm),
                ln(self.attn_prenorm_x_dst), mode=mode)
            if self.has_pos_emb:
                gamma, beta = self.attn_prenorm_r.weight, self.attn_prenorm_r.bias
                packs['pos'] = ((self.to_k_r.weight * gamma.unsqueeze(0)).contiguous(),
                                (self.to_v_r.weight * gamma.unsqueeze(0)).contiguous(),
                                (self.to_v_r.weight @ beta + self.to_v_r.bias).contiguous())
                packs['pos_modes'] = ops.pack_attn_modes(packs['pos'][0], packs['pos'][1], mode=mode)
        self._packs, self._pack_key = packs, key
        return packs

    def forward(self,
                x: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]],
                r: Union[None, torch.Tensor, NormalizedEdgeFeatures],
                edge_index: Union[torch.Tensor, CSRGraph],
                source_cache: Optional[dict] = None,
                pre: Optional[Tuple[torch.Tensor, ...]] = None,
                next_layer: Optional['AttentionLayer'] = None):
        """``r``: the relation's positional embedding, one row per edge of ``edge_index`` in its order (for a
        mode-replicated CSRGraph: one row per *base* edge), either raw (reference semantics) or already
        row-normalised (``NormalizedEdgeFeatures``).
        ``source_cache`` (bipartite calls only; not in the reference): a dict owned by the caller in which the layer keeps
        the key / value projections of ``x[0]``.  A caller that applies this layer several times to the SAME source rows
        (the decoder's recurrent proposal steps, modules/qcnet_decoder.py:219-227) passes the same dict and the projections
        are computed once; the caller guarantees ``x[0]`` and the layer's weights do not change while it holds the dict.
        ``next_layer`` / ``pre`` (not in the reference) chain a stack of layers in which this layer's output is the next
        one's destination input (x[1], or x of a layer over one node set): with ``next_layer`` the call returns
        ``(out, pre_next)`` where ``pre_next`` holds that layer's LayerNorm + to_q (bipartite) or to_q / to_k / to_v projections
        of ``out``, computed while the row was in registers; the caller hands it to that layer as ``pre`` (together with
        the very tensor ``out`` as its destination input) and the layer launches no projection of its destination rows."""
        if self.training and self.dropout > 0:
            raise RuntimeError('AttentionLayer is inference-only: call .eval() (dropout is not implemented)')
        packs = self._kernel_packs()
        mode = packs['mode']
        single = isinstance(x, torch.Tensor)   # one node set: source and destination share LN(x) (reference :74-75)
        if single:
            x_res = x.contiguous()
            n_src = n_dst = x_res.size(0)
        else:
            x_src, x_res = x[0].contiguous(), x[1].contiguous()
            n_src, n_dst = x_src.size(0), x_res.size(0)
        if isinstance(edge_index, CSRGraph):
            graph = edge_index
            if graph.n_src != n_src or graph.n_dst != n_dst:
                raise ValueError('CSRGraph is %d -> %d nodes but x has %d -> %d rows'
                                 % (graph.n_src, graph.n_dst, n_src, n_dst))
        else:
            graph = CSRGraph.from_edge_index(edge_index, n_src, n_dst)
        if self.has_pos_emb and r is not None:
            if not isinstance(r, NormalizedEdgeFeatures):
                r = NormalizedEdgeFeatures.from_raw(r, self.attn_prenorm_r.eps)
            if r.num_rows != graph.num_feature_rows:
                raise ValueError('r has %d rows for %d edges' % (r.num_rows, graph.num_feature_rows))
        else:
            r = None
        # pre: LayerNorm + q/k/v/s projections (one or two launches)
        if not single:
            ln_s, ln_d = self.attn_prenorm_x_src, self.attn_prenorm_x_dst
            # (the cached projections are tagged by the source tensor's storage, version counter and shape and by the
            # layer's parameter versions: a reused dict with other sources, or after a weight update, recomputes)
            src_tag = (x_src.data_ptr(), x_src._version, tuple(x_src.shape), self._pack_key)
            kept_kv = source_cache.get(self) if source_cache is not None else None
            if kept_kv is None or kept_kv[0] != src_tag:
                kv, _ = ops.ln_linear(x_src, ln_s.weight, ln_s.bias, *packs['pre_src'], mode=mode)
                if source_cache is not None:
                    source_cache[self] = (src_tag, kv, x_src)      # x_src kept alive: its address is part of the tag
            else:
                kv = kept_kv[1]
            k, v = kv
            if pre is not None:
                (q,), s = pre, None
            elif CHAIN_NODE_SIDE:
                (q,), _ = ops.ln_linear(x_res, ln_d.weight, ln_d.bias, *packs['pre_q'], mode=mode)
                s = None
            else:
                (q, s), _ = ops.ln_linear(x_res, ln_d.weight, ln_d.bias, *packs['pre_dst'], mode=mode)
        else:
            ln = self.attn_prenorm_x_src
            if pre is not None:
                (q, k, v), s = pre, None
            elif CHAIN_NODE_SIDE:
                (q, k, v), _ = ops.ln_linear(x_res, ln.weight, ln.bias, *packs['pre_qkv'], mode=mode)
                s = None
            else:
                (q, k, v, s), _ = ops.ln_linear(x_res, ln.weight, ln.bias, *packs['pre_all'], mode=mode)
        if q.size(0) != n_dst:
            raise ValueError('pre holds %d destination rows, x has %d' % (q.size(0), n_dst))
        # propagate: gather + segment softmax + weighted scatter-add (one launch)
        if r is not None:
            wk, wv, bv = packs['pos']
            if self.uses_attn_modes(graph, n_dst, n_src, r.num_rows):
                # mode-replicated relation: per-edge projections once per (row, edge) on the matrix cores — bf16 operands
                # in the bf16 mode (BASELINE.json configs[3]), split f16 (fp32-equivalent) otherwise
                r_op = r.bf16 if mode == ops.MATMUL_BF16 else r.split
                # (not in the bf16 mode: there a projection is one MFMA on a 256-byte row and recomputing it is cheaper than
                # reading 1 KiB of tables per edge — cfg4: 92.6 scenarios/s recomputing, 89.9 from tables)
                if (source_cache is not None and CACHE_EDGE_PROJECTIONS and mode != ops.MATMUL_BF16
                        and _tables_fit(graph, q.device)):
                    # the caller applies this layer to the same sources again (source_cache contract): the per-edge
                    # projections k[j] + W_kr LN_r(r_e), v[j] + W_vr LN_r(r_e) are the same every time — the first call
                    # stores them in CSR-slot order, the later ones read them back instead of recomputing
                    tag = (graph.rowptr.data_ptr(), graph.col.data_ptr(), graph.num_base_edges, r_op.data_ptr(), k.data_ptr(), mode)
                    kept = source_cache.get(('edge_kv', self))
                    if kept is not None and kept[0] == tag:
                        agg = ops.attn_modes_cached(q, kept[1], kept[2], graph.rowptr, bv, graph.rep, graph.layout,
                                                    graph.num_base_edges)
                    else:
                        try:
                            agg, kk_e, vv_e = ops.attn_modes_store(q, k, v, r_op, graph.rowptr, graph.col, graph.eid,
                                                                   packs['pos_modes'], bv, graph.rep, graph.layout, graph.col_shift)
                            source_cache[('edge_kv', self)] = (tag, kk_e, vv_e)
                        except torch.cuda.OutOfMemoryError:      # no room for the tables: recompute
                            source_cache.pop(('edge_kv', self), None)
                            agg = ops.attn_modes(q, k, v, r_op, graph.rowptr, graph.col, graph.eid, packs['pos_modes'], bv,
                                                 graph.rep, graph.layout, graph.col_shift)
                else:
                    agg = ops.attn_modes(q, k, v, r_op, graph.rowptr, graph.col, graph.eid, packs['pos_modes'], bv,
                                         graph.rep, graph.layout, graph.col_shift)
            else:
                agg = ops.attn_fused(q, k, v, r.dense, graph.rowptr, graph.col, graph.eid, wk, wv, bv, graph.rep,
                                     graph.layout,
                                     graph.col_shift, graph.row_index, graph.src_offset)
        else:
            if graph.rep != 1 or graph.col_shift != 0 or graph.row_index is not None or graph.src_offset is not None:
                raise ValueError('mode-replicated / windowed graphs need a positional embedding')
            agg = ops.attn_fwd(q, k, v, None, None, graph.rowptr, graph.col, graph.eid)
        # post: gate/update, to_out, post-norm residual, feed-forward block (one launch)
        # the gate's LN(x_dst) is recomputed inside the kernel (xn_dst = None): cheaper than a [N, 128] round trip
        if s is not None:
            out = ops.attn_post(agg, None, s, x_res, packs['post'], mode=mode)
            return out if next_layer is None else (out, next_layer.project_destinations(out))
        if next_layer is None:
            return ops.attn_post_chain(agg, x_res, packs['post_chain'], mode=mode)[0]
        nwt, nbias, nln = next_layer._destination_projection()
        outs = ops.attn_post_chain(agg, x_res, packs['post_chain'], nwt, nbias, nln.weight, nln.bias, mode=mode,
                                   next_operand_scale=next_layer._kernel_packs()['dst_operand_scale'])
        return outs[0], tuple(outs[1:])

    def _destination_projection(self):
        """(weights, biases, LayerNorm) of what this layer applies to its destination rows before the propagate: to_q behind
        attn_prenorm_x_dst (bipartite), to_q / to_k / to_v behind the shared LayerNorm (one node set)."""
        packs = self._kernel_packs()
        if self.bipartite:
            return packs['pre_q'] + (self.attn_prenorm_x_dst,)
        return packs['pre_qkv'] + (self.attn_prenorm_x_src,)

    def project_destinations(self, x_dst: torch.Tensor) -> Tuple[torch.Tensor, ...]:
        """``pre`` of :meth:`forward` for destination rows ``x_dst`` as a launch of its own (the unchained form)."""
        wt, bias, ln = self._destination_projection()
        ys, _ = ops.ln_linear(x_dst.contiguous(), ln.weight, ln.bias, wt, bias, mode=self._kernel_packs()['mode'])
        return tuple(ys)

    @staticmethod
    def edge_table_bytes(graph: CSRGraph) -> int:
        """Bytes of the two projection tables of one (layer, relation): 1 KiB per base edge plus the row padding."""
        return 2 * 512 * ops.attn_modes_table_slots(graph.num_base_edges, graph.rowptr.numel() - 1)

    @staticmethod
    def uses_attn_modes(graph: CSRGraph, n_dst: int, n_src: int, n_feat_rows: int) -> bool:
        """Dispatch of the propagate: mode-replicated relations (2..6 queries per CSR row) go to qcn_attn_modes_* unless a
        row count reaches 2^23 — that kernel addresses rows by 32-bit byte offsets and would refuse; qcn_attn_fused_f32
        handles the same graph at any size."""
        return (USE_ATTN_MODES and 1 < graph.rep <= 6 and graph.layout != ops.LAYOUT_PLAIN
                and ops.attn_modes_supported(n_dst, n_src, n_feat_rows))
