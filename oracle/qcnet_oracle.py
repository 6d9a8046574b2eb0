"""CPU ORACLE — test infrastructure, never the product.

A plain-PyTorch (CPU, fp32) restatement of the algorithm the reference runs on the QCNet /
TempEnsLearnAgg forward path, written as pure functions over a ``state_dict`` whose keys are the
reference's own parameter names.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the product package never does.

Pinning status (see DESIGN.md "Oracle"):
* ``fourier_embedding``, ``mlp_layer``, ``angle_between_2d_vectors``, ``wrap_angle`` are checked in
  ``tests/test_oracle_pinned.py`` against golden vectors produced by importing the reference's own
  ``layers/fourier_embedding.py``, ``layers/mlp_layer.py``, ``utils/geometry.py`` in the build
  container (``oracle/make_golden.py``) — PINNED.
* Everything that needs ``torch_geometric`` 2.3.0 / ``torch_cluster`` 1.6.1 (attention propagate +
  softmax, radius / radius_graph, dense_to_sparse, subgraph, coalesce, Batch collation, and therefore
  the module-level functions) restates the published semantics of those pinned versions, which are
  not installed anywhere in this environment and for which the reference holds no tests or
  fixtures: PARITY UNPINNED for those rows.

Conventions: ``sd`` = state_dict (CPU tensors), ``p`` = key prefix ending in '.', ``hp`` = hyper-parameter
dict with the reference's ctor argument names (``predictors/qcnet_sliding.py:58-87``).
"""
import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# geometry  (utils/geometry.py)
# --------------------------------------------------------------------------------------------
def angle_between_2d_vectors(ctr_vector: Tensor, nbr_vector: Tensor) -> Tensor:
    """utils/geometry.py:19-23 — atan2(cross, dot) of two 2-D vectors."""
    cross = ctr_vector[..., 0] * nbr_vector[..., 1] - ctr_vector[..., 1] * nbr_vector[..., 0]
    dot = (ctr_vector[..., :2] * nbr_vector[..., :2]).sum(dim=-1)
    return torch.atan2(cross, dot)


def wrap_angle(angle: Tensor, min_val: float = -math.pi, max_val: float = math.pi) -> Tensor:
    """utils/geometry.py:47-51 — floor-mod wrap into [min_val, max_val)."""
    return min_val + (angle + max_val) % (max_val - min_val)


def rel_geometry(pos_src: Tensor, pos_dst: Tensor, head_src: Tensor, head_dst: Tensor,
                 edge_index: Tensor, extra: Optional[Tensor] = None) -> Tensor:
    """The r-vector every relation builds: [‖Δp‖, angle(heading_dst, Δp), wrap(Δheading), (extra)].

    Pattern of modules/qcnet_map_encoder.py:121-128, modules/qcnet_agent_encoder.py:133-139,161-166,171-176,
    modules/qcnet_decoder.py:161-167,183-188,201-206.  Δ = src − dst.
    """
    src, dst = edge_index[0], edge_index[1]
    rel_pos = pos_src[src] - pos_dst[dst]
    rel_head = wrap_angle(head_src[src] - head_dst[dst])
    head_vec_dst = torch.stack([head_dst.cos(), head_dst.sin()], dim=-1)
    cols = [torch.norm(rel_pos[:, :2], p=2, dim=-1),
            angle_between_2d_vectors(head_vec_dst[dst], rel_pos[:, :2]),
            rel_head]
    if extra is not None:
        cols.append(extra.to(rel_head.dtype))
    return torch.stack(cols, dim=-1)


# --------------------------------------------------------------------------------------------
# third-party graph utilities restated (torch_cluster 1.6.1, torch_geometric 2.3.0)
# --------------------------------------------------------------------------------------------
def _ptr_from_batch(batch: Optional[Tensor], n: int, batch_size: int) -> Tensor:
    """torch_cluster radius wrapper: ptr = bucketize(arange(batch_size + 1), batch) (sorted batch)."""
    if batch is None:
        return torch.tensor([0, n], dtype=torch.long)
    return torch.bucketize(torch.arange(batch_size + 1, dtype=torch.long), batch.cpu())


def radius(x: Tensor, y: Tensor, r: float, batch_x: Optional[Tensor] = None, batch_y: Optional[Tensor] = None,
           max_num_neighbors: int = 32) -> Tensor:
    """torch_cluster.radius restated (call sites modules/qcnet_agent_encoder.py:158, modules/qcnet_decoder.py:174).

    Canonical order (SURVEY.md §8 G1 = upstream device kernel): for every y row, ascending, the x rows of
    the same example in ascending index order with fp32 ``fma(dy, dy, dx*dx) < r*r``; at most
    ``max_num_neighbors`` per y row (lowest x indices kept).  Returns ``[2, E]`` int64, row0 = y index,
    row1 = x index.  The distance predicate is evaluated in float64 here and cross-checked bit-for-bit
    against the C restatement ``oracle/radius_ref.c`` (true fp32 ``fmaf``) in the tests.
    """
    if x.numel() == 0 or y.numel() == 0:
        return torch.empty(2, 0, dtype=torch.long)
    x = x.reshape(-1, 1) if x.dim() == 1 else x
    y = y.reshape(-1, 1) if y.dim() == 1 else y
    batch_size = 1
    if batch_x is not None:
        batch_size = int(batch_x.max()) + 1
    if batch_y is not None:
        batch_size = max(batch_size, int(batch_y.max()) + 1)
    ptr_x = _ptr_from_batch(batch_x if batch_size > 1 else None, x.size(0), batch_size).tolist()
    ptr_y = _ptr_from_batch(batch_y if batch_size > 1 else None, y.size(0), batch_size).tolist()
    r2 = torch.tensor(float(r) * float(r), dtype=torch.float64).float()  # C++ wrapper passes (scalar_t)(r*r)
    rows: List[Tensor] = []
    cols: List[Tensor] = []
    for b in range(batch_size):
        xs = x[ptr_x[b]:ptr_x[b + 1]].float()
        ys = y[ptr_y[b]:ptr_y[b + 1]].float()
        if xs.size(0) == 0 or ys.size(0) == 0:
            continue
        # fp32 chain: d = dx*dx (rounded), then d = fma(dim_k, dim_k, d) for the remaining dims.
        diff = (xs[None, :, :] - ys[:, None, :])                      # fp32 subtract, exact restatement
        d = (diff[..., 0].double() * diff[..., 0].double()).float()  # product of two fp32 is exact in fp64
        for k in range(1, x.size(1)):
            # fma: exact product + d, rounded once to fp32.  fp64 holds the 48-bit product exactly; the
            # fp64 add may round, which differs from a true fmaf only on ties no sane input hits — the
            # C restatement is the bit-level authority and the tests compare the two.
            d = (diff[..., k].double() * diff[..., k].double() + d.double()).float()
        hit = d < r2                                                  # [ny, nx]
        keep = hit & (hit.long().cumsum(dim=1) <= max_num_neighbors)
        yy, xx = keep.nonzero(as_tuple=True)                          # row-major: y asc, x asc
        rows.append(yy + ptr_y[b])
        cols.append(xx + ptr_x[b])
    if not rows:
        return torch.empty(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(rows), torch.cat(cols)], dim=0)


def radius_graph(x: Tensor, r: float, batch: Optional[Tensor] = None, loop: bool = False,
                 max_num_neighbors: int = 32, flow: str = 'source_to_target') -> Tensor:
    """torch_cluster.radius_graph restated (call sites modules/qcnet_map_encoder.py:141,
    modules/qcnet_agent_encoder.py:168, modules/qcnet_decoder.py:194): radius(x, x) with cap+1 when
    ``loop=False``, rows swapped so row0 = neighbour (source), row1 = query (target), self pairs dropped."""
    assert flow in ('source_to_target', 'target_to_source')
    ei = radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    if flow == 'source_to_target':
        row, col = ei[1], ei[0]
    else:
        row, col = ei[0], ei[1]
    if not loop:
        keep = row != col
        row, col = row[keep], col[keep]
    return torch.stack([row, col], dim=0)


def dense_to_sparse(adj: Tensor) -> Tensor:
    """torch_geometric.utils.dense_to_sparse for a [B, n, n] mask (modules/qcnet_agent_encoder.py:130,
    modules/qcnet_decoder.py:213): row-major nonzero, src = b*n + i, dst = b*n + j."""
    b, i, j = adj.nonzero(as_tuple=True)
    return torch.stack([b * adj.size(1) + i, b * adj.size(2) + j], dim=0)


def bipartite_dense_to_sparse(adj: Tensor) -> Tensor:
    """utils/graph.py:88-94 — [B, n, m] mask → src = b*n + i, dst = b*m + j (row-major nonzero order)."""
    idx = adj.nonzero(as_tuple=True)
    if len(idx) == 3:
        return torch.stack([idx[0] * adj.size(1) + idx[1], idx[0] * adj.size(2) + idx[2]], dim=0)
    return torch.stack(idx, dim=0)


def subgraph(subset: Tensor, edge_index: Tensor) -> Tensor:
    """torch_geometric.utils.subgraph with a bool node mask, no relabelling
    (modules/qcnet_agent_encoder.py:170): keep edges whose both endpoints are in the mask, order kept."""
    keep = subset[edge_index[0]] & subset[edge_index[1]]
    return edge_index[:, keep]


def coalesce_max(edge_index: Tensor, edge_attr: Tensor) -> Tuple[Tensor, Tensor]:
    """torch_geometric.utils.coalesce(reduce='max') as used by merge_edges (utils/graph.py:45-54):
    sort by key src*N + dst with N = max node id + 1, drop duplicates, attribute = max over duplicates."""
    if edge_index.size(1) == 0:
        return edge_index, edge_attr
    n = int(edge_index.max()) + 1
    key = edge_index[0] * n + edge_index[1]
    uniq, inverse = torch.unique(key, sorted=True, return_inverse=True)
    out_attr = torch.zeros(uniq.numel(), dtype=torch.long)
    out_attr.scatter_reduce_(0, inverse, edge_attr.long(), reduce='amax', include_self=False)
    out_index = torch.stack([uniq // n, uniq % n], dim=0)
    return out_index, out_attr.to(edge_attr.dtype)


def merge_edges(edge_indices: Sequence[Tensor], edge_attrs: Sequence[Tensor]) -> Tuple[Tensor, Tensor]:
    """utils/graph.py:45-54 with reduce='max' (the only call, modules/qcnet_map_encoder.py:146)."""
    return coalesce_max(torch.cat(list(edge_indices), dim=1), torch.cat(list(edge_attrs), dim=0))


def segment_softmax(src: Tensor, index: Tensor, num_nodes: Optional[int] = None) -> Tensor:
    """torch_geometric.utils.softmax(src, index, ptr=None): per-destination
    exp(x - max) / (sum + 1e-16), N = index.max() + 1 (call: layers/attention_layer.py:98)."""
    if num_nodes is None:
        num_nodes = int(index.max()) + 1 if index.numel() > 0 else 0
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    seg_max = src.new_zeros((num_nodes,) + src.shape[1:])
    seg_max.scatter_reduce_(0, idx, src, reduce='amax', include_self=False)
    out = (src - seg_max.index_select(0, index)).exp()
    seg_sum = src.new_zeros((num_nodes,) + src.shape[1:]).scatter_add_(0, idx, out) + 1e-16
    return out / seg_sum.index_select(0, index)


# --------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------
# bf16-operand mode (BASELINE.json configs[3] "bf16 projections"; no counterpart in the reference's files, whose Linear
# layers are plain fp32 — its counterpart would be autocast).  Inside ``with bf16_operands():`` every nn.Linear of the path
# multiplies operands rounded to bf16 (round to nearest even, as torch's .bfloat16()) and accumulates in fp32 — the
# arithmetic contract of the product's QCN_MATMUL_BF16 mode (include/qcnet_hip.h), restated here so that the HIP path can be
# checked against the SAME operands:
#   * biases, the x column of a FourierEmbedding's first layer (a rank-1 fp32 term), LayerNorm, activations, residuals and
#     the GRU recurrence stay fp32;
#   * the positional projections to_k_r / to_v_r of the decoder's mode relations (t2m / pl2m / a2m) take the row-normalised
#     r and the weight with the attn_prenorm_r scale folded in as their bf16 operands (W * gamma), the shift as an fp32 bias
#     (W beta + b); those of the encoder relations stay fp32 (the product folds them onto the queries in fp32).
_BF16_OPERANDS = False


class bf16_operands:
    def __enter__(self):
        global _BF16_OPERANDS
        self._prev, _BF16_OPERANDS = _BF16_OPERANDS, True
        return self

    def __exit__(self, *exc):
        global _BF16_OPERANDS
        _BF16_OPERANDS = self._prev
        return False


def _rb(x: Tensor) -> Tensor:
    return x.bfloat16().float()


def _linear(sd: Mapping[str, Tensor], p: str, x: Tensor) -> Tensor:
    if _BF16_OPERANDS:
        return F.linear(_rb(x), _rb(sd[p + 'weight']), sd.get(p + 'bias'))
    return F.linear(x, sd[p + 'weight'], sd.get(p + 'bias'))


def _layer_norm(sd: Mapping[str, Tensor], p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.size(-1),), sd[p + 'weight'], sd[p + 'bias'], 1e-5)


def fourier_embedding(sd: Mapping[str, Tensor], p: str, continuous_inputs: Optional[Tensor],
                      categorical_embs: Optional[List[Tensor]] = None) -> Tensor:
    """layers/fourier_embedding.py:49-67."""
    if continuous_inputs is None:
        if categorical_embs is None:
            raise ValueError('Both continuous_inputs and categorical_embs are None')
        x = torch.stack(categorical_embs).sum(dim=0)
    else:
        freqs = sd[p + 'freqs.weight']                                  # [in, F]
        ang = continuous_inputs.unsqueeze(-1) * freqs * 2 * math.pi      # :58  ((x*w)*2)*pi
        feat = torch.cat([ang.cos(), ang.sin(), continuous_inputs.unsqueeze(-1)], dim=-1)  # :60
        x = None
        for i in range(freqs.size(0)):                                   # :62-64
            if _BF16_OPERANDS:   # the 2F Fourier features are matrix operands, the x column a rank-1 fp32 term
                w, nf = sd['%smlps.%d.0.weight' % (p, i)], feat.size(-1) - 1
                h = (F.linear(_rb(feat[:, i, :nf]), _rb(w[:, :nf]), sd['%smlps.%d.0.bias' % (p, i)])
                     + feat[:, i, nf:] * w[:, nf])
            else:
                h = _linear(sd, '%smlps.%d.0.' % (p, i), feat[:, i])
            h = F.relu(_layer_norm(sd, '%smlps.%d.1.' % (p, i), h))
            h = _linear(sd, '%smlps.%d.3.' % (p, i), h)
            x = h if x is None else x + h
        if categorical_embs is not None:                                 # :65-66
            x = x + torch.stack(categorical_embs).sum(dim=0)
    return _linear(sd, p + 'to_out.2.', F.relu(_layer_norm(sd, p + 'to_out.0.', x)))  # :67


def mlp_layer(sd: Mapping[str, Tensor], p: str, x: Tensor) -> Tensor:
    """layers/mlp_layer.py:27-36 — Linear, LayerNorm, ReLU, Linear."""
    h = F.relu(_layer_norm(sd, p + 'mlp.1.', _linear(sd, p + 'mlp.0.', x)))
    return _linear(sd, p + 'mlp.3.', h)


def attention_layer(sd: Mapping[str, Tensor], p: str,
                    x: Union[Tensor, Tuple[Tensor, Tensor]], r: Optional[Tensor], edge_index: Tensor,
                    num_heads: int = 8, head_dim: int = 16) -> Tensor:
    """layers/attention_layer.py:70-122 with PyG's propagate/softmax written out.

    ``has_pos_emb`` is read off the state_dict (presence of ``to_k_r``); non-bipartite layers carry the
    aliased ``attn_prenorm_x_dst`` keys with identical values (:60-61), so both branches read the same names.
    """
    has_pos_emb = (p + 'to_k_r.weight') in sd
    if isinstance(x, torch.Tensor):                                      # :74-75
        x_src = x_dst = _layer_norm(sd, p + 'attn_prenorm_x_src.', x)
    else:                                                                # :76-80
        x_src = _layer_norm(sd, p + 'attn_prenorm_x_src.', x[0])
        x_dst = _layer_norm(sd, p + 'attn_prenorm_x_dst.', x[1])
        x = x[1]
    mode_relation = any(tag in p for tag in ('t2m_', 'pl2m_', 'a2m_'))
    r_hat = None
    if has_pos_emb and r is not None:                                    # :81-82
        if _BF16_OPERANDS and mode_relation:
            r_hat = F.layer_norm(r, (r.size(-1),), None, None, 1e-5)
        r = _layer_norm(sd, p + 'attn_prenorm_r.', r)
    # _attn_block :110-119
    H, D = num_heads, head_dim
    q = _linear(sd, p + 'to_q.', x_dst).view(-1, H, D)
    k = _linear(sd, p + 'to_k.', x_src).view(-1, H, D)
    v = _linear(sd, p + 'to_v.', x_src).view(-1, H, D)
    src, dst = edge_index[0], edge_index[1]
    q_i, k_j, v_j = q.index_select(0, dst), k.index_select(0, src), v.index_select(0, src)   # PyG __collect__
    if has_pos_emb and r is not None and r_hat is not None:              # bf16 mode, decoder mode relations (see above)
        gamma, beta = sd[p + 'attn_prenorm_r.weight'], sd[p + 'attn_prenorm_r.bias']
        wk, wv = sd[p + 'to_k_r.weight'] * gamma, sd[p + 'to_v_r.weight'] * gamma
        bv = sd[p + 'to_v_r.weight'] @ beta + sd[p + 'to_v_r.bias']
        k_j = k_j + (F.linear(_rb(r_hat), _rb(wk)) + sd[p + 'to_k_r.weight'] @ beta).view(-1, H, D)
        v_j = v_j + F.linear(_rb(r_hat), _rb(wv), bv).view(-1, H, D)
    elif has_pos_emb and r is not None:                                  # message :94-96
        fp32 = _BF16_OPERANDS                                            # bf16 mode: encoder relations stay fp32
        lin = (lambda q_, x_: F.linear(x_, sd[q_ + 'weight'], sd.get(q_ + 'bias'))) if fp32 else \
            (lambda q_, x_: _linear(sd, q_, x_))
        k_j = k_j + lin(p + 'to_k_r.', r).view(-1, H, D)
        v_j = v_j + lin(p + 'to_v_r.', r).view(-1, H, D)
    sim = (q_i * k_j).sum(dim=-1) * (D ** -0.5)                          # :97
    attn = segment_softmax(sim, dst)                                     # :98
    msg = v_j * attn.unsqueeze(-1)                                       # :101
    agg = q.new_zeros(q.size(0), H, D).index_add_(0, dst, msg)           # PyG aggregate (scatter add, dim_size = |dst nodes|)
    inputs = agg.view(-1, H * D)                                         # update :106-108
    g = torch.sigmoid(_linear(sd, p + 'to_g.', torch.cat([inputs, x_dst], dim=-1)))
    upd = inputs + g * (_linear(sd, p + 'to_s.', x_dst) - inputs)
    x = x + _layer_norm(sd, p + 'attn_postnorm.', _linear(sd, p + 'to_out.', upd))      # :83
    ff = _linear(sd, p + 'ff_mlp.3.', F.relu(_linear(sd, p + 'ff_mlp.0.', _layer_norm(sd, p + 'ff_prenorm.', x))))
    return x + _layer_norm(sd, p + 'ff_postnorm.', ff)                   # :84


def gru_last_hidden(sd: Mapping[str, Tensor], p: str, seq: Tensor, h0: Tensor) -> Tensor:
    """nn.GRU(1 layer, batch_first=False) final hidden state (modules/qcnet_decoder.py:86-88,257).
    Gate order r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1 - z) * n + z * h."""
    w_ih, w_hh = sd[p + 'weight_ih_l0'], sd[p + 'weight_hh_l0']
    b_ih, b_hh = sd[p + 'bias_ih_l0'], sd[p + 'bias_hh_l0']
    h = h0
    gi_all = F.linear(_rb(seq), _rb(w_ih), b_ih) if _BF16_OPERANDS else F.linear(seq, w_ih, b_ih)   # recurrence: fp32
    for t in range(seq.size(0)):
        gi = gi_all[t]
        gh = F.linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi.chunk(3, dim=-1)
        h_r, h_z, h_n = gh.chunk(3, dim=-1)
        rg = torch.sigmoid(i_r + h_r)
        zg = torch.sigmoid(i_z + h_z)
        ng = torch.tanh(i_n + rg * h_n)
        h = (1.0 - zg) * ng + zg * h
    return h


# --------------------------------------------------------------------------------------------
# modules
# --------------------------------------------------------------------------------------------
def _is_batch(data) -> bool:
    """Stands in for ``isinstance(data, Batch)`` (modules/qcnet_agent_encoder.py:148)."""
    return hasattr(data, 'num_graphs')


def map_encoder(sd: Mapping[str, Tensor], p: str, data, hp: Mapping[str, object]) -> Dict[str, Tensor]:
    """modules/qcnet_map_encoder.py:95-173 (argoverse_v2; input_dim 2 or 3)."""
    assert hp['input_dim'] in (2, 3)
    C, T = hp['hidden_dim'], hp['num_historical_steps']
    H, D = hp['num_heads'], hp['head_dim']
    pos_pt = data['map_point']['position'][:, :2]
    orient_pt = data['map_point']['orientation']
    pos_pl = data['map_polygon']['position'][:, :2]
    orient_pl = data['map_polygon']['orientation']

    x_pt_cat = [sd[p + 'type_pt_emb.weight'][data['map_point']['type'].long()],          # :111-112
                sd[p + 'side_pt_emb.weight'][data['map_point']['side'].long()]]
    x_pl_cat = [sd[p + 'type_pl_emb.weight'][data['map_polygon']['type'].long()],        # :113-114
                sd[p + 'int_pl_emb.weight'][data['map_polygon']['is_intersection'].long()]]
    dim3 = hp.get('input_dim', 2) == 3                                                   # :103-108
    if dim3:
        x_pt_in = torch.stack([data['map_point']['magnitude'], data['map_point']['height']], dim=-1)
        x_pl_in = data['map_polygon']['height'].unsqueeze(-1)
    else:
        x_pt_in, x_pl_in = data['map_point']['magnitude'].unsqueeze(-1), None
    x_pt = fourier_embedding(sd, p + 'x_pt_emb.', x_pt_in, x_pt_cat)
    x_pl = fourier_embedding(sd, p + 'x_pl_emb.', x_pl_in, x_pl_cat)

    def with_dz(r, ei, src_store):   # input_dim == 3: [dist, angle, dz, d_orient] (:129-135,156-162)
        if not dim3:
            return r
        dz = data[src_store]['position'][ei[0], 2] - data['map_polygon']['position'][ei[1], 2]
        return torch.stack([r[:, 0], r[:, 1], dz, r[:, 2]], dim=-1)

    ei_pt2pl = data['map_point', 'to', 'map_polygon']['edge_index']                      # :120-138
    r_pt2pl = with_dz(rel_geometry(pos_pt, pos_pl, orient_pt, orient_pl, ei_pt2pl), ei_pt2pl, 'map_point')
    r_pt2pl = fourier_embedding(sd, p + 'r_pt2pl_emb.', r_pt2pl, None)

    ei_data = data['map_polygon', 'to', 'map_polygon']['edge_index']                     # :140-147
    ei_rad = radius_graph(pos_pl, hp['pl2pl_radius'],
                          data['map_polygon']['batch'] if _is_batch(data) else None,
                          loop=False, max_num_neighbors=300)
    type_data = data['map_polygon', 'to', 'map_polygon']['type']
    ei_pl2pl, type_pl2pl = merge_edges([ei_rad, ei_data],
                                       [type_data.new_zeros(ei_rad.size(1), dtype=torch.uint8), type_data])
    r_pl2pl = with_dz(rel_geometry(pos_pl, pos_pl, orient_pl, orient_pl, ei_pl2pl), ei_pl2pl, 'map_polygon')   # :148-162
    r_pl2pl = fourier_embedding(sd, p + 'r_pl2pl_emb.', r_pl2pl,
                                [sd[p + 'type_pl2pl_emb.weight'][type_pl2pl.long()]])    # :165

    for i in range(hp['num_map_layers']):                                                # :167-169
        x_pl = attention_layer(sd, '%spt2pl_layers.%d.' % (p, i), (x_pt, x_pl), r_pt2pl, ei_pt2pl, H, D)
        x_pl = attention_layer(sd, '%spl2pl_layers.%d.' % (p, i), x_pl, r_pl2pl, ei_pl2pl, H, D)
    x_pl = x_pl.repeat_interleave(T, dim=0).reshape(-1, T, C)                            # :170-171
    return {'x_pt': x_pt, 'x_pl': x_pl, 'edge_index_pl2pl': ei_pl2pl, 'type_pl2pl': type_pl2pl}


def agent_encoder(sd: Mapping[str, Tensor], p: str, data, map_enc: Mapping[str, Tensor], time_shift: int,
                  hp: Mapping[str, object]) -> Dict[str, Tensor]:
    """modules/qcnet_agent_encoder.py:93-189."""
    C, T = hp['hidden_dim'], hp['num_historical_steps']
    H, D = hp['num_heads'], hp['head_dim']
    span = hp['time_span'] if hp.get('time_span') is not None else T
    ts = time_shift
    A = data['agent']['num_nodes']
    mask = data['agent']['valid_mask'][:, ts:T + ts]                                     # :97
    pos_a = data['agent']['position'][:, ts:T + ts, :2]
    motion = torch.cat([pos_a.new_zeros(A, 1, 2), pos_a[:, 1:] - pos_a[:, :-1]], dim=1)  # :99-100
    head_a = data['agent']['heading'][:, ts:T + ts]
    head_vec_a = torch.stack([head_a.cos(), head_a.sin()], dim=-1)
    pos_pl = data['map_polygon']['position'][:, :2]
    orient_pl = data['map_polygon']['orientation']
    vel = data['agent']['velocity'][:, ts:T + ts, :2]
    type_emb = sd[p + 'type_a_emb.weight'][data['agent']['type'].long()].repeat_interleave(T, dim=0)  # :109

    x_a = torch.stack([torch.norm(motion, p=2, dim=-1),                                  # :116-120
                       angle_between_2d_vectors(head_vec_a, motion),
                       torch.norm(vel, p=2, dim=-1),
                       angle_between_2d_vectors(head_vec_a, vel)], dim=-1)
    x_a = fourier_embedding(sd, p + 'x_a_emb.', x_a.reshape(-1, 4), [type_emb])          # :123

    pos_t = pos_a.reshape(-1, 2)                                                         # agent-major (a*T + t)
    head_t = head_a.reshape(-1)
    ei_t = dense_to_sparse(mask.unsqueeze(2) & mask.unsqueeze(1))                        # :129-132
    ei_t = ei_t[:, ei_t[1] > ei_t[0]]
    ei_t = ei_t[:, ei_t[1] - ei_t[0] <= span]
    r_t = rel_geometry(pos_t, pos_t, head_t, head_t, ei_t, extra=ei_t[0] - ei_t[1])      # :133-139
    r_t = fourier_embedding(sd, p + 'r_t_emb.', r_t, None)

    pos_s = pos_a.transpose(0, 1).reshape(-1, 2)                                         # time-major (t*A + a) :142-145
    head_s = head_a.transpose(0, 1).reshape(-1)
    mask_s = mask.transpose(0, 1).reshape(-1)
    pos_pl_rep = pos_pl.repeat(T, 1)                                                     # :146-147
    orient_pl_rep = orient_pl.repeat(T)
    if _is_batch(data):                                                                  # :148-157
        batch_s = torch.cat([data['agent']['batch'] + data.num_graphs * t for t in range(T)], dim=0)
        batch_pl = torch.cat([data['map_polygon']['batch'] + data.num_graphs * t for t in range(T)], dim=0)
    else:
        batch_s = torch.arange(T).repeat_interleave(A)
        batch_pl = torch.arange(T).repeat_interleave(data['map_polygon']['num_nodes'])
    ei_pl2a = radius(pos_s, pos_pl_rep, hp['pl2a_radius'], batch_s, batch_pl, max_num_neighbors=300)  # :158
    ei_pl2a = ei_pl2a[:, mask_s[ei_pl2a[1]]]                                             # :160
    r_pl2a = rel_geometry(pos_pl_rep, pos_s, orient_pl_rep, head_s, ei_pl2a)             # :161-166
    r_pl2a = fourier_embedding(sd, p + 'r_pl2a_emb.', r_pl2a, None)
    ei_a2a = radius_graph(pos_s, hp['a2a_radius'], batch_s, loop=False, max_num_neighbors=300)        # :168
    ei_a2a = subgraph(mask_s, ei_a2a)                                                    # :170
    r_a2a = rel_geometry(pos_s, pos_s, head_s, head_s, ei_a2a)                           # :171-176
    r_a2a = fourier_embedding(sd, p + 'r_a2a_emb.', r_a2a, None)

    x_pl_s = map_enc['x_pl'].transpose(0, 1).reshape(-1, C)                              # :184
    for i in range(hp['num_agent_layers']):                                              # :179-187
        x_a = x_a.reshape(-1, C)
        x_a = attention_layer(sd, '%st_attn_layers.%d.' % (p, i), x_a, r_t, ei_t, H, D)
        x_a = x_a.reshape(-1, T, C).transpose(0, 1).reshape(-1, C)
        x_a = attention_layer(sd, '%spl2a_attn_layers.%d.' % (p, i), (x_pl_s, x_a), r_pl2a, ei_pl2a, H, D)
        x_a = attention_layer(sd, '%sa2a_attn_layers.%d.' % (p, i), x_a, r_a2a, ei_a2a, H, D)
        x_a = x_a.reshape(T, -1, C).transpose(0, 1)
    return {'x_a': x_a, 'edge_index_t': ei_t, 'edge_index_pl2a': ei_pl2a, 'edge_index_a2a': ei_a2a}


def encoder(sd, p, data, time_shift, hp):
    """modules/qcnet_encoder.py:69-72."""
    m = map_encoder(sd, p + 'map_encoder.', data, hp)
    a = agent_encoder(sd, p + 'agent_encoder.', data, m, time_shift, hp)
    return {**m, **a}


def decoder(sd: Mapping[str, Tensor], p: str, data, scene_enc: Mapping[str, Tensor], time_shift: int,
            hp: Mapping[str, object], temporal_m: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """modules/qcnet_decoder.py:141-294; with ``temporal_m`` given, modules/temporal_aggregate.py:141-301
    (identical except the mode-query sum of :261-263 and the 'final_m' key of :299).
    Both ``output_head`` branches (predictors/qcnet_sliding.py:424 defaults it to False)."""
    C, T, K = hp['hidden_dim'], hp['num_historical_steps'], hp['num_modes']
    Tf, R, OD = hp['num_future_steps'], hp['num_recurrent_steps'], hp['output_dim']
    H, D, L = hp['num_heads'], hp['head_dim'], hp['num_dec_layers']
    t2m_steps = hp['num_t2m_steps'] if hp.get('num_t2m_steps') is not None else T
    ts = time_shift
    A = data['agent']['num_nodes']
    P = data['map_polygon']['num_nodes']
    pos_m = data['agent']['position'][:, T - 1 + ts, :2]                                 # :145-147
    head_m = data['agent']['heading'][:, T - 1 + ts]

    x_t = scene_enc['x_a'].reshape(-1, C)                                                # :149-152
    x_pl = scene_enc['x_pl'][:, T - 1].repeat(K, 1)
    x_a = scene_enc['x_a'][:, -1].repeat(K, 1)
    m = sd[p + 'mode_emb.weight'].repeat(A, 1)

    mask_src = data['agent']['valid_mask'][:, ts:T + ts].clone()                         # :154-156
    mask_src[:, :T - t2m_steps] = False
    mask_dst = data['agent']['predict_mask'].any(dim=-1, keepdim=True).repeat(1, K)

    pos_t = data['agent']['position'][:, ts:T + ts, :2].reshape(-1, 2)                   # :158-170
    head_t = data['agent']['heading'][:, ts:T + ts].reshape(-1)
    ei_t2m_1 = bipartite_dense_to_sparse(mask_src.unsqueeze(2) & mask_dst[:, -1:].unsqueeze(1))
    r_t2m = rel_geometry(pos_t, pos_m, head_t, head_m, ei_t2m_1, extra=(ei_t2m_1[0] % T) - T + 1)
    r_t2m = fourier_embedding(sd, p + 'r_t2m_emb.', r_t2m, None)
    ei_t2m = bipartite_dense_to_sparse(mask_src.unsqueeze(2) & mask_dst.unsqueeze(1))
    r_t2m = r_t2m.repeat_interleave(K, dim=0)

    pos_pl = data['map_polygon']['position'][:, :2]                                      # :172-192
    orient_pl = data['map_polygon']['orientation']
    batch_a = data['agent']['batch'] if _is_batch(data) else None
    batch_pl = data['map_polygon']['batch'] if _is_batch(data) else None
    ei_pl2m_1 = radius(pos_m, pos_pl, hp['pl2m_radius'], batch_a, batch_pl, max_num_neighbors=300)
    ei_pl2m_1 = ei_pl2m_1[:, mask_dst[ei_pl2m_1[1], 0]]
    r_pl2m = rel_geometry(pos_pl, pos_m, orient_pl, head_m, ei_pl2m_1)
    r_pl2m = fourier_embedding(sd, p + 'r_pl2m_emb.', r_pl2m, None)
    ei_pl2m = torch.cat([ei_pl2m_1 + i * ei_pl2m_1.new_tensor([[P], [A]]) for i in range(K)], dim=1)
    r_pl2m = r_pl2m.repeat(K, 1)

    ei_a2m_1 = radius_graph(pos_m, hp['a2m_radius'], batch_a, loop=False, max_num_neighbors=300)   # :194-211
    ei_a2m_1 = ei_a2m_1[:, mask_src[:, -1][ei_a2m_1[0]] & mask_dst[ei_a2m_1[1], 0]]
    r_a2m = rel_geometry(pos_m, pos_m, head_m, head_m, ei_a2m_1)
    r_a2m = fourier_embedding(sd, p + 'r_a2m_emb.', r_a2m, None)
    ei_a2m = torch.cat([ei_a2m_1 + i * A for i in range(K)], dim=1)
    r_a2m = r_a2m.repeat(K, 1)

    ei_m2m = dense_to_sparse(mask_dst.unsqueeze(2) & mask_dst.unsqueeze(1))              # :213

    def cross_stack(m: Tensor, stage: str) -> Tensor:
        for i in range(L):                                                               # :220-226 / :258-263
            m = m.reshape(-1, C)
            m = attention_layer(sd, '%st2m_%s_attn_layers.%d.' % (p, stage, i), (x_t, m), r_t2m, ei_t2m, H, D)
            m = m.reshape(-1, K, C).transpose(0, 1).reshape(-1, C)
            m = attention_layer(sd, '%spl2m_%s_attn_layers.%d.' % (p, stage, i), (x_pl, m), r_pl2m, ei_pl2m, H, D)
            m = attention_layer(sd, '%sa2m_%s_attn_layers.%d.' % (p, stage, i), (x_a, m), r_a2m, ei_a2m, H, D)
            m = m.reshape(K, -1, C).transpose(0, 1).reshape(-1, C)
        m = attention_layer(sd, '%sm2m_%s_attn_layer.' % (p, stage), m, None, ei_m2m, H, D)   # :227 / :264
        return m.reshape(-1, K, C)

    out_head = bool(hp.get('output_head', False))
    locs, scales, hlocs, hconcs = [], [], [], []
    for _ in range(R):                                                                   # :219-233
        m = cross_stack(m, 'propose')
        locs.append(mlp_layer(sd, p + 'to_loc_propose_pos.', m))
        scales.append(mlp_layer(sd, p + 'to_scale_propose_pos.', m))
        if out_head:
            hlocs.append(mlp_layer(sd, p + 'to_loc_propose_head.', m))
            hconcs.append(mlp_layer(sd, p + 'to_conc_propose_head.', m))
    loc_propose_pos = torch.cumsum(torch.cat(locs, dim=-1).view(-1, K, Tf, OD), dim=-2)  # :234-242
    scale_propose_pos = torch.cumsum(F.elu(torch.cat(scales, dim=-1).view(-1, K, Tf, OD), alpha=1.0) + 1.0,
                                     dim=-2) + 0.1
    if out_head:                                                                         # :243-250
        loc_propose_head = torch.cumsum(torch.tanh(torch.cat(hlocs, dim=-1).unsqueeze(-1)) * math.pi, dim=-2)
        conc_propose_head = 1.0 / (torch.cumsum(F.elu(torch.cat(hconcs, dim=-1).unsqueeze(-1)) + 1.0, dim=-2) + 0.02)
        m = fourier_embedding(sd, p + 'y_emb.', torch.cat([loc_propose_pos, wrap_angle(loc_propose_head)],
                                                          dim=-1).view(-1, OD + 1), None)
    else:
        loc_propose_head = loc_propose_pos.new_zeros(A, K, Tf, 1)                        # :251-254
        conc_propose_head = scale_propose_pos.new_zeros(A, K, Tf, 1)
        m = fourier_embedding(sd, p + 'y_emb.', loc_propose_pos.reshape(-1, OD), None)   # :255
    m = m.reshape(-1, Tf, C).transpose(0, 1)                                             # :256
    h0 = sd[p + 'traj_emb_h0'].repeat(m.size(1), 1)                                      # :257
    m = gru_last_hidden(sd, p + 'traj_emb.', m, h0)
    if temporal_m is not None:                                                           # temporal_aggregate.py:261-263
        m = torch.cat([m.reshape(-1, 1, K, C), temporal_m], dim=1).sum(dim=1).reshape(-1, C)
    m = cross_stack(m, 'refine')                                                         # :258-265
    loc_refine_pos = mlp_layer(sd, p + 'to_loc_refine_pos.', m).view(-1, K, Tf, OD) + loc_propose_pos   # :266-267
    scale_refine_pos = F.elu(mlp_layer(sd, p + 'to_scale_refine_pos.', m).view(-1, K, Tf, OD), alpha=1.0) + 1.0 + 0.1
    pi = mlp_layer(sd, p + 'to_pi.', m).squeeze(-1)                                      # :280
    if out_head:                                                                         # :271-274
        loc_refine_head = torch.tanh(mlp_layer(sd, p + 'to_loc_refine_head.', m).unsqueeze(-1)) * math.pi + loc_propose_head
        conc_refine_head = 1.0 / (F.elu(mlp_layer(sd, p + 'to_conc_refine_head.', m).unsqueeze(-1)) + 1.0 + 0.02)
    else:
        loc_refine_head = loc_refine_pos.new_zeros(A, K, Tf, 1)
        conc_refine_head = scale_refine_pos.new_zeros(A, K, Tf, 1)
    return {
        'loc_propose_pos': loc_propose_pos, 'scale_propose_pos': scale_propose_pos,
        'loc_propose_head': loc_propose_head, 'conc_propose_head': conc_propose_head,
        'loc_refine_pos': loc_refine_pos, 'scale_refine_pos': scale_refine_pos,
        'loc_refine_head': loc_refine_head, 'conc_refine_head': conc_refine_head,
        'pi': pi, ('final_m' if temporal_m is not None else 'm'): m,
        'edge_index_pl2m_clone': ei_pl2m_1,
        'edge_index_t2m': ei_t2m, 'edge_index_pl2m': ei_pl2m, 'edge_index_a2m': ei_a2m, 'edge_index_m2m': ei_m2m,
    }


# --------------------------------------------------------------------------------------------
# predictors
# --------------------------------------------------------------------------------------------
def qcnet_forward(sd, data, time_shift: int, hp, p: str = ''):
    """predictors/qcnet_sliding.py:174-177."""
    scene_enc = encoder(sd, p + 'encoder.', data, time_shift, hp)
    pred = decoder(sd, p + 'decoder.', data, scene_enc, time_shift, hp)
    return pred, scene_enc


def temp_ens_learn_agg_forward(sd, data, hp, num_shifts: int = 10):
    """predictors/temp_ens_learn_agg.py:196-221: ten sliding QCNet forwards, stack pred['m'], TemporalAggregate on
    the last frame's scene encoding; trajectories truncated to the first 50 future steps (:219-220) and
    pi soft-maxed (:261)."""
    temporal_m = []
    scene_enc = None
    for ts in range(1, num_shifts + 1):
        pred, scene_enc = qcnet_forward(sd, data, ts, hp, p='base_model.')
        temporal_m.append(pred['m'])
    temporal_m = torch.stack(temporal_m, dim=1)
    pred = decoder(sd, 'temporal_aggregate_layer.', data, scene_enc, num_shifts, hp, temporal_m=temporal_m)
    od = hp['output_dim']
    for stage in ('propose', 'refine'):                                                  # :204-220
        cols = [pred['loc_%s_pos' % stage][..., :od], pred['scale_%s_pos' % stage][..., :od]]
        if hp.get('output_head', False):
            cols = [cols[0], pred['loc_%s_head' % stage], cols[1], pred['conc_%s_head' % stage]]
        pred['traj_' + stage] = torch.cat(cols, dim=-1)[..., :50, :]
    pred['pi_softmax'] = F.softmax(pred['pi'], dim=-1)
    return pred


# --------------------------------------------------------------------------------------------
# PyG-formulation CPU baseline for one relation: FourierEmbedding + AttentionLayer
# --------------------------------------------------------------------------------------------
def relation_forward(sd_emb, sd_attn, x_src, x_dst, r_raw, edge_index):
    r = fourier_embedding(sd_emb, '', r_raw, None)
    return attention_layer(sd_attn, '', (x_src, x_dst), r, edge_index)
