"""AttentionLayer — drop-in for the reference's PyG ``MessagePassing`` block (layers/attention_layer.py:24-122).

Same constructor arguments, same ``forward(x, r, edge_index)`` contract and the same ``state_dict`` keys
(including the aliased ``attn_prenorm_x_dst`` of non-bipartite layers, reference :60-61), but no PyG: the
gather → segment softmax → scatter-add of ``propagate`` is one HIP kernel over a destination-sorted CSR
(``qcn_attn_fwd_f32``).  Inference only — dropout sites (:49,53,99) are identities in eval mode and the layer
refuses to run in training mode with ``dropout > 0``.
"""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops
from ..utils.graph import CSRGraph
from ..utils.weight_init import weight_init


class NormalizedEdgeFeatures:
    """Positional embeddings of a relation after the layer-independent half of ``attn_prenorm_r``:
    ``rhat = (r - mean) / sqrt(var + eps)`` per edge, [E_base, 128].  Every AttentionLayer that consumes the
    relation applies only its own affine (folded into its projection weights), so the normalisation is done
    once per relation instead of once per layer (reference: layers/attention_layer.py:82 in every call)."""

    __slots__ = ('rhat',)

    def __init__(self, rhat: torch.Tensor) -> None:
        self.rhat = rhat

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
        self._fold_key = None
        self._fold = None

    def _folded_pos_weights(self):
        """wk = to_k_r.weight * gamma, wv = to_v_r.weight * gamma, bv = to_v_r.weight @ beta + to_v_r.bias with
        (gamma, beta) the affine of attn_prenorm_r: what qcn_attn_fused_f32 consumes.  Cached per parameter version."""
        prm = (self.to_k_r.weight, self.to_v_r.weight, self.to_v_r.bias, self.attn_prenorm_r.weight,
               self.attn_prenorm_r.bias)
        key = tuple((p.data_ptr(), p._version) for p in prm)
        if key != self._fold_key:
            with torch.no_grad():
                gamma, beta = self.attn_prenorm_r.weight, self.attn_prenorm_r.bias
                wk = (self.to_k_r.weight * gamma.unsqueeze(0)).contiguous()
                wv = (self.to_v_r.weight * gamma.unsqueeze(0)).contiguous()
                bv = (self.to_v_r.weight @ beta + self.to_v_r.bias).contiguous()
            self._fold, self._fold_key = (wk, wv, bv), key
        return self._fold

    def forward(self,
                x: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]],
                r: Union[None, torch.Tensor, NormalizedEdgeFeatures],
                edge_index: Union[torch.Tensor, CSRGraph]) -> torch.Tensor:
        """``r``: the relation's positional embedding, one row per edge of ``edge_index`` in its order (for a
        mode-replicated CSRGraph: one row per *base* edge), either raw (reference semantics) or already
        row-normalised (``NormalizedEdgeFeatures``)."""
        if self.training and self.dropout > 0:
            raise RuntimeError('AttentionLayer is inference-only: call .eval() (dropout is not implemented)')
        if isinstance(x, torch.Tensor):
            x_src = x_dst = self.attn_prenorm_x_src(x)
        else:
            x_src, x_dst = x
            x_src = self.attn_prenorm_x_src(x_src)
            x_dst = self.attn_prenorm_x_dst(x_dst)
            x = x[1]
        if isinstance(edge_index, CSRGraph):
            graph = edge_index
            if graph.n_src != x_src.size(0) or graph.n_dst != x_dst.size(0):
                raise ValueError('CSRGraph is %d -> %d nodes but x has %d -> %d rows'
                                 % (graph.n_src, graph.n_dst, x_src.size(0), x_dst.size(0)))
        else:
            graph = CSRGraph.from_edge_index(edge_index, x_src.size(0), x_dst.size(0))
        if self.has_pos_emb and r is not None:
            if not isinstance(r, NormalizedEdgeFeatures):
                r = NormalizedEdgeFeatures.from_raw(r, self.attn_prenorm_r.eps)
            if r.rhat.size(0) != graph.num_base_edges:
                raise ValueError('r has %d rows for %d edges' % (r.rhat.size(0), graph.num_base_edges))
        else:
            r = None
        x = x + self.attn_postnorm(self._attn_block(x_src, x_dst, r, graph))
        x = x + self.ff_postnorm(self.ff_mlp(self.ff_prenorm(x)))
        return x

    def _attn_block(self, x_src: torch.Tensor, x_dst: torch.Tensor, r: Optional[NormalizedEdgeFeatures],
                    graph: CSRGraph) -> torch.Tensor:
        q = self.to_q(x_dst)
        k = self.to_k(x_src)
        v = self.to_v(x_src)
        if r is not None:
            wk, wv, bv = self._folded_pos_weights()
            agg = ops.attn_fused(q, k, v, r.rhat, graph.rowptr, graph.col, graph.eid, wk, wv, bv, graph.rep,
                                 graph.layout)
        else:
            if graph.rep != 1:
                raise ValueError('mode-replicated graphs need a positional embedding')
            agg = ops.attn_fwd(q, k, v, None, None, graph.rowptr, graph.col, graph.eid)
        g = torch.sigmoid(self.to_g(torch.cat([agg, x_dst], dim=-1)))
        return self.to_out(agg + g * (self.to_s(x_dst) - agg))
