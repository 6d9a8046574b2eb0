"""QCNetAgentEncoder — drop-in for modules/qcnet_agent_encoder.py:32-189 (same ctor kwargs,
forward(data, map_enc, time_shift) and state_dict keys).

Graphs: the temporal graph keeps the reference's dense_to_sparse order; polygon→agent and agent→agent edges come
from the HIP radius kernel with the validity filters of reference :160,170 fused in; every relation is converted
to a destination-sorted CSR once and reused by both layers.  The T-fold tiling of the polygon features
(reference :184) is folded away: sources are gathered from the single [P, C] copy.
"""
import collections
from typing import Dict, Mapping, Optional

import torch
import torch.nn as nn

from .. import ops
from ..layers import attention_layer
from ..layers import AttentionLayer, FourierEmbedding, NormalizedEdgeFeatures
from ..utils.geometry import angle_between_2d_vectors
from ..utils.graph import CSRGraph, mask_outer_edges
from ..utils.weight_init import weight_init
from ..utils.prep import run_prep
from ._common import node_ptr, rel_geometry


class QCNetAgentEncoder(nn.Module):

    def __init__(self,
                 dataset: str,
                 input_dim: int,
                 hidden_dim: int,
                 num_historical_steps: int,
                 time_span: Optional[int],
                 pl2a_radius: float,
                 a2a_radius: float,
                 num_freq_bands: int,
                 num_layers: int,
                 num_heads: int,
                 head_dim: int,
                 dropout: float) -> None:
        super().__init__()
        if dataset != 'argoverse_v2':
            raise ValueError('{} is not a valid dataset'.format(dataset))
        self.dataset = dataset
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_historical_steps = num_historical_steps
        self.time_span = time_span if time_span is not None else num_historical_steps
        self.pl2a_radius = pl2a_radius
        self.a2a_radius = a2a_radius
        self.num_freq_bands = num_freq_bands
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.dropout = dropout

        self.type_a_emb = nn.Embedding(10, hidden_dim)
        for name, dim in (('x_a_emb', 4), ('r_t_emb', 4), ('r_pl2a_emb', 3), ('r_a2a_emb', 3)):
            setattr(self, name, FourierEmbedding(input_dim=dim, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands))
        layer = dict(hidden_dim=hidden_dim, num_heads=num_heads, head_dim=head_dim, dropout=dropout, has_pos_emb=True)
        self.t_attn_layers = nn.ModuleList([AttentionLayer(bipartite=False, **layer) for _ in range(num_layers)])
        self.pl2a_attn_layers = nn.ModuleList([AttentionLayer(bipartite=True, **layer) for _ in range(num_layers)])
        self.a2a_attn_layers = nn.ModuleList([AttentionLayer(bipartite=False, **layer) for _ in range(num_layers)])
        self.apply(weight_init)

    # ------------------------------------------------------------------------------------------------------------
    # shared pieces
    # ------------------------------------------------------------------------------------------------------------
    def _agent_features(self, pos: torch.Tensor, prev: torch.Tensor, head: torch.Tensor, vel: torch.Tensor) -> torch.Tensor:
        """x_a inputs (reference :116-120): [|motion|, angle(heading, motion), |velocity|, angle(heading, velocity)]."""
        motion = pos - prev
        head_vec = torch.stack([head.cos(), head.sin()], dim=-1)
        return torch.stack([torch.norm(motion, p=2, dim=-1),
                            angle_between_2d_vectors(ctr_vector=head_vec, nbr_vector=motion),
                            torch.norm(vel, p=2, dim=-1),
                            angle_between_2d_vectors(ctr_vector=head_vec, nbr_vector=vel)], dim=-1)

    def _temporal_edges(self, mask: torch.Tensor) -> torch.Tensor:
        """dst later than src by at most time_span steps, both valid (reference :129-132); node id a * n + t."""
        return mask_outer_edges(mask, mask, 1, self.time_span)     # one enumeration kernel, no dense [A, n, n] adjacency

    def _step_ptrs(self, data, n_steps: int, dev):
        """Example boundaries for the spatial searches: one example per (time step, scenario), time-major node ids
        (reference batch ids b + B * t, :148-157)."""
        A, P = int(data['agent']['num_nodes']), int(data['map_polygon']['num_nodes'])
        steps = torch.arange(n_steps, device=dev, dtype=torch.int64).unsqueeze(1)
        ptr_a, ptr_p = node_ptr(data, 'agent', dev), node_ptr(data, 'map_polygon', dev)
        ptr_s = torch.cat([(steps * A + ptr_a[:-1]).reshape(-1), ptr_a.new_tensor([n_steps * A])])
        ptr_pl = torch.cat([(steps * P + ptr_p[:-1]).reshape(-1), ptr_p.new_tensor([n_steps * P])])
        return ptr_s, ptr_pl

    def _spatial_relations(self, data, pos_a, head_a, mask, n_steps: int):
        """Polygon→agent and agent→agent relations over ``n_steps`` time steps (time-major node ids t * A + a):
        edge lists from the HIP radius kernel with the validity filters of reference :160,170 fused in,
        destination-sorted CSR, row-normalised positional embeddings."""
        A, P = int(data['agent']['num_nodes']), int(data['map_polygon']['num_nodes'])
        dev = pos_a.device
        pos_pl = data['map_polygon']['position'][:, :2].contiguous()
        orient_pl = data['map_polygon']['orientation'].contiguous()
        pos_s = pos_a.transpose(0, 1).reshape(-1, 2).contiguous()
        head_s = head_a.transpose(0, 1).reshape(-1)
        mask_s = mask.transpose(0, 1).reshape(-1).contiguous()
        pos_pl_rep = pos_pl.repeat(n_steps, 1)
        orient_pl_rep = orient_pl.repeat(n_steps)
        ptr_s, ptr_pl = self._step_ptrs(data, n_steps, dev)

        ei_pl2a, off_pl2a = ops.radius_pairs(pos_s, pos_pl_rep, self.pl2a_radius, ptr_s, ptr_pl, 300, mask_x=mask_s)
        g_pl2a = CSRGraph.from_edge_index(ei_pl2a, n_steps * P, n_steps * A, validate=False).fold_sources(P)
        geo_pl2a = rel_geometry(pos_pl_rep, pos_s, orient_pl_rep, head_s, ei_pl2a)
        r_pl2a = NormalizedEdgeFeatures.deferred(lambda: self.r_pl2a_emb(
            continuous_inputs=geo_pl2a, categorical_embs=None, normalized_output=True), (geo_pl2a,))

        ei_a2a, off_a2a = ops.radius_pairs(pos_s, pos_s, self.a2a_radius, ptr_s, ptr_s, 300 + 1, drop_self=True,
                                           mask_x=mask_s, mask_y=mask_s, y_first=False)
        g_a2a = CSRGraph.from_sorted(ei_a2a, off_a2a, n_steps * A, n_steps * A)
        geo_a2a = rel_geometry(pos_s, pos_s, head_s, head_s, ei_a2a)
        r_a2a = NormalizedEdgeFeatures.deferred(lambda: self.r_a2a_emb(
            continuous_inputs=geo_a2a, categorical_embs=None, normalized_output=True), (geo_a2a,))
        return dict(g_pl2a=g_pl2a, r_pl2a=r_pl2a, off_pl2a=off_pl2a, g_a2a=g_a2a, r_a2a=r_a2a, off_a2a=off_a2a)

    def _layers(self, x_a, x_pl, g_t, r_t, g_pl2a, r_pl2a, g_a2a, r_a2a) -> torch.Tensor:
        T, C = self.num_historical_steps, self.hidden_dim
        for i in range(self.num_layers):
            x_a = x_a.reshape(-1, C)
            x_a = self.t_attn_layers[i](x_a, r_t, g_t)
            x_a = x_a.reshape(-1, T, C).transpose(0, 1).reshape(-1, C)
            x_a = self._pl2a_a2a(i, x_pl, x_a, r_pl2a, g_pl2a, r_a2a, g_a2a)
            x_a = x_a.reshape(T, -1, C).transpose(0, 1)
        return x_a

    def _pl2a_a2a(self, i: int, x_pl, x_a, r_pl2a, g_pl2a, r_a2a, g_a2a, pre=None) -> torch.Tensor:
        """Layer i's polygon->agent and agent->agent attention over the same (time-major) rows, chained: the first one's
        node-side kernel also applies the second one's LayerNorm + to_q / to_k / to_v (layers/attention_layer.py)."""
        if not attention_layer.CHAIN_NODE_SIDE:
            return self.a2a_attn_layers[i](self.pl2a_attn_layers[i]((x_pl, x_a), r_pl2a, g_pl2a), r_a2a, g_a2a)
        x_a, qkv = self.pl2a_attn_layers[i]((x_pl, x_a), r_pl2a, g_pl2a, pre=pre, next_layer=self.a2a_attn_layers[i])
        return self.a2a_attn_layers[i](x_a, r_a2a, g_a2a, pre=qkv)

    # ------------------------------------------------------------------------------------------------------------
    # sliding-window reuse: everything that depends on ABSOLUTE time only, computed once for all frames
    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def precompute(self, data, first_shift: int, last_shift: int) -> Dict[str, object]:
        """See :meth:`_precompute` (run on the preparation stream, utils/prep.py)."""
        dev = data['agent']['position'].device
        return run_prep(dev, lambda: self._precompute(data, first_shift, last_shift))

    def _precompute(self, data, first_shift: int, last_shift: int) -> Dict[str, object]:
        """The sliding loop (predictors/temp_ens_learn_agg.py:197) encodes windows [ts, ts + T) for consecutive
        ``ts``.  A node (agent a, absolute step tau) has the same input features, the same spatial neighbours and the
        same relative geometry in every window that contains it (only the very first step of a window differs: its
        motion vector is zero, reference :99), so the Fourier embeddings, the radius searches and the CSR of the
        spatial relations are computed once over tau in [first_shift, last_shift + T) and every frame takes views."""
        T, C = self.num_historical_steps, self.hidden_dim
        s0, n_frames = int(first_shift), int(last_shift) - int(first_shift) + 1
        n_abs = T + n_frames - 1
        agent = data['agent']
        A = int(agent['num_nodes'])
        mask = agent['valid_mask'][:, s0:s0 + n_abs].contiguous()
        pos = agent['position'][:, s0:s0 + n_abs, :2].contiguous()
        head = agent['heading'][:, s0:s0 + n_abs].contiguous()
        vel = agent['velocity'][:, s0:s0 + n_abs, :2].contiguous()
        if s0 >= 1:
            prev = agent['position'][:, s0 - 1:s0 + n_abs - 1, :2]
        else:
            prev = torch.cat([pos[:, :1], pos[:, :-1]], dim=1)
        type_emb = [self.type_a_emb(agent['type'].long())]
        x_main = self.x_a_emb(continuous_inputs=self._agent_features(pos, prev, head, vel).view(-1, 4),
                              categorical_embs=type_emb, categorical_repeat=n_abs).view(A, n_abs, C)
        pf, hf, vf = pos[:, :n_frames], head[:, :n_frames], vel[:, :n_frames]
        x_first = self.x_a_emb(continuous_inputs=self._agent_features(pf, pf, hf, vf).reshape(-1, 4),
                               categorical_embs=type_emb, categorical_repeat=n_frames).view(A, n_frames, C)

        ei_t = self._temporal_edges(mask)                                    # node id a * n_abs + u
        pos_t, head_t = pos.reshape(-1, 2), head.reshape(-1)
        geo_t = rel_geometry(pos_t, pos_t, head_t, head_t, ei_t, extra=ei_t[0] - ei_t[1])
        r_t = NormalizedEdgeFeatures.deferred(lambda: self.r_t_emb(
            continuous_inputs=geo_t, categorical_embs=None, normalized_output=True), (geo_t,))
        agent_of_edge = torch.div(ei_t[0], n_abs, rounding_mode='floor')
        cache = dict(first_shift=s0, n_frames=n_frames, n_abs=n_abs, x_main=x_main, x_first=x_first, r_t=r_t,
                     t_agent=agent_of_edge, t_src=ei_t[0] - agent_of_edge * n_abs, t_dst=ei_t[1] - agent_of_edge * n_abs)
        cache.update(self._spatial_relations(data, pos, head, mask, n_abs))
        return cache

    def _forward_cached(self, data, map_enc, time_shift: int, cache, return_edge_index: bool) -> Dict[str, torch.Tensor]:
        T = self.num_historical_steps
        A, P = int(data['agent']['num_nodes']), int(data['map_polygon']['num_nodes'])
        u0 = int(time_shift) - cache['first_shift']
        if not 0 <= u0 < cache['n_frames']:
            raise ValueError('time_shift %d is outside the precomputed range' % time_shift)
        x_a = torch.cat([cache['x_first'][:, u0:u0 + 1], cache['x_main'][:, u0 + 1:u0 + T]], dim=1)

        keep = (cache['t_src'] >= u0) & (cache['t_dst'] < u0 + T)
        sel = keep.nonzero(as_tuple=True)[0]
        base = cache['t_agent'][sel] * T - u0
        ei_t = torch.stack([base + cache['t_src'][sel], base + cache['t_dst'][sel]], dim=0)
        g_t = CSRGraph.from_edge_index(ei_t, A * T, A * T, validate=False)
        g_t.eid = sel.to(torch.int32)[g_t.eid.long()]
        g_t.num_feature_rows = int(cache['t_src'].numel())

        g_pl2a = cache['g_pl2a'].window(u0 * A, (u0 + T) * A, n_src=P)
        g_a2a = cache['g_a2a'].window(u0 * A, (u0 + T) * A, n_src=T * A, col_shift=u0 * A)
        x_a = self._layers(x_a, map_enc['x_pl'][:, 0], g_t, cache['r_t'], g_pl2a, cache['r_pl2a'], g_a2a, cache['r_a2a'])
        out = {'x_a': x_a}
        if return_edge_index:   # the reference-order COO lists of this frame (costs host round trips)
            lo, hi = int(cache['off_pl2a'][u0 * P]), int(cache['off_pl2a'][(u0 + T) * P])
            shift = cache['g_pl2a'].edge_index.new_tensor([[u0 * P], [u0 * A]])
            out['edge_index_pl2a'] = cache['g_pl2a'].edge_index[:, lo:hi] - shift
            lo, hi = int(cache['off_a2a'][u0 * A]), int(cache['off_a2a'][(u0 + T) * A])
            out['edge_index_a2a'] = cache['g_a2a'].edge_index[:, lo:hi] - u0 * A
            out['edge_index_t'] = ei_t
        return out

    @torch.no_grad()
    def forward_frames(self, data, map_enc: Mapping[str, torch.Tensor], cache) -> torch.Tensor:
        """All frames of the precomputed range in ONE pass: returns ``x_a`` [F * A, T, C] with frame f (time shift
        first_shift + f) in rows [f * A, (f + 1) * A) — row for row what ``forward(data, map_enc, first_shift + f)``
        gives.  Every layer is launched once over the F-fold destination set: the spatial relations are frame groups of
        the absolute-time CSR (``CSRGraph.frames``), the temporal relation is one edge list over all frames."""
        T, C = self.num_historical_steps, self.hidden_dim
        A, P = int(data['agent']['num_nodes']), int(data['map_polygon']['num_nodes'])
        F = cache['n_frames']
        dev = cache['x_main'].device
        # frame f, step t -> absolute step f + t (its first step takes the zero-motion variant)
        steps = torch.arange(F, device=dev).unsqueeze(1) + torch.arange(1, T, device=dev).unsqueeze(0)     # [F, T-1]
        x_a = torch.cat([cache['x_first'].transpose(0, 1).unsqueeze(2),
                         cache['x_main'][:, steps].permute(1, 0, 2, 3)], dim=2)                             # [F, A, T, C]

        f_ids = torch.arange(F, device=dev).unsqueeze(1)
        keep = (cache['t_src'].unsqueeze(0) >= f_ids) & (cache['t_dst'].unsqueeze(0) < f_ids + T)          # [F, E_abs]
        f_sel, e_sel = keep.nonzero(as_tuple=True)
        base = (f_sel * A + cache['t_agent'][e_sel]) * T - f_sel
        ei_t = torch.stack([base + cache['t_src'][e_sel], base + cache['t_dst'][e_sel]], dim=0)
        g_t = CSRGraph.from_edge_index(ei_t, F * A * T, F * A * T, validate=False)
        g_t.eid = e_sel.to(torch.int32)[g_t.eid.long()]
        g_t.num_feature_rows = int(cache['t_src'].numel())

        g_pl2a = cache['g_pl2a'].frames(F, T * A, A, n_src=P)
        g_a2a = cache['g_a2a'].frames(F, T * A, A, n_src=F * T * A, col_shift=A, src_stride=T * A)
        x_pl = map_enc['x_pl'][:, 0]
        for i in range(self.num_layers):
            x_a = x_a.reshape(-1, C)                                                     # (f, a, t)
            x_a = self.t_attn_layers[i](x_a, cache['r_t'], g_t)
            x_a = x_a.view(F, A, T, C).transpose(1, 2).reshape(-1, C)                    # (f, t, a)
            x_a = self._pl2a_a2a(i, x_pl, x_a, cache['r_pl2a'], g_pl2a, cache['r_a2a'], g_a2a)
            x_a = x_a.view(F, T, A, C).transpose(1, 2)                                   # (f, a, t)
        return x_a.reshape(F * A, T, C)

    def streaming_plan(self, needed_from: int):
        """Which (frame-relative) steps of a later frame cannot be taken from the long run.

        The temporal attention is causal with a reach of ``time_span`` steps, the other two relations do not mix time,
        and the only frame-dependent input is the zero motion vector of a window's first step (reference :99).  So the
        output of layer l at offset d = u - f inside frame f >= 1 equals the long run's value at absolute step u as soon
        as d > l * time_span.  ``D[l]`` lists the offsets of layer l's output that are (a) needed — by the consumer
        (offsets >= ``needed_from``) for the last layer, as a source or as the destination's own row by ``D[l + 1]``
        otherwise — and (b) not yet equal to the long run."""
        S, NL, T = self.time_span, self.num_layers, self.num_historical_steps
        special = lambda l, d: d <= l * S
        D = [None] * (NL + 1)
        D[NL] = [d for d in range(int(needed_from), T) if special(NL, d)]
        for l in range(NL, 0, -1):
            need = set()
            for d in D[l]:
                need.update(dp for dp in range(max(0, d - S), d) if special(l - 1, dp))
                if special(l - 1, d):
                    need.add(d)
            D[l - 1] = sorted(need)
        return D

    @torch.no_grad()
    def stream_frames(self, data, map_enc: Mapping[str, torch.Tensor], cache, needed_from: int):
        """The agent encodings of all frames like :meth:`forward_frames`, but only steps >= ``needed_from`` of every
        frame are produced (the decoder reads the last num_t2m_steps steps, modules/qcnet_decoder.py:143,151), as
        ``(table [N, C], index [F, A, T - needed_from])`` with ``x_a[f * A + a, needed_from + j] = table[index[f, a, j]]``,
        and the work the frames share is done once:

        * one LONG RUN of the layers over the absolute steps [0, T + F - 1) of all agents (frame 0 is its prefix),
        * per later frame only the few steps of :meth:`streaming_plan` (for the reference configuration one step per
          layer: offsets 0 / 10 / 20), appended to the long run's rows as extra destinations of the same launches.
        Row for row the arithmetic is that of ``forward(data, map_enc, first_shift + f)``."""
        T, C, S, NL = self.num_historical_steps, self.hidden_dim, self.time_span, self.num_layers
        A, P = int(data['agent']['num_nodes']), int(data['map_polygon']['num_nodes'])
        F, n_abs = cache['n_frames'], cache['n_abs']
        dev = cache['x_main'].device
        NA = A * n_abs
        plan = run_prep(dev, lambda: self._stream_plan_graphs(cache, A, P, needed_from))
        x_pl = map_enc['x_pl'][:, 0]
        z = plan['z0']
        for l in range(1, NL + 1):
            L = plan['layers'][l - 1]
            # input rows: [long run | the special destinations' own rows | special rows of the previous layer]; the
            # first NA + n_own output rows are the ones that go on
            inp = torch.cat([z[:NA], z[L['own_idx']], z[NA:]]) if L['n_own'] else z[:NA]
            # (the rows that go on are a prefix of the temporal layer's output: so are their chained projections)
            n_on = NA + L['n_own']
            if attention_layer.CHAIN_NODE_SIDE:
                y, (q,) = self.t_attn_layers[l - 1](inp, cache['r_t'], L['g_t'], next_layer=self.pl2a_attn_layers[l - 1])
                pre = (q[:n_on],)
            else:
                y, pre = self.t_attn_layers[l - 1](inp, cache['r_t'], L['g_t']), None
            z = self._pl2a_a2a(l - 1, x_pl, y[:n_on], cache['r_pl2a'], L['g_pl2a'], cache['r_a2a'], L['g_a2a'], pre=pre)
        return z, plan['index']          # the last layer's output already is [long run | special rows]

    def _stream_plan_graphs(self, cache, A: int, P: int, needed_from: int):
        """The index work of :meth:`stream_frames` (depends on the validity masks only): per layer the temporal graph
        over [long run | special destinations | previous special rows], the gather of the special destinations' own
        rows and the spatial relations with their per-destination indirections; the layer-0 rows; the final index."""
        T, C, S, NL = self.num_historical_steps, self.hidden_dim, self.time_span, self.num_layers
        F, n_abs = cache['n_frames'], cache['n_abs']
        dev = cache['x_main'].device
        D = self.streaming_plan(needed_from)
        if any(d > 0 for d in D[0]):
            raise AssertionError('layer-0 features differ from the long run at the first step only')
        NA = A * n_abs
        i64 = dict(device=dev, dtype=torch.int64)
        shape = self._stream_shape_plan(D, A, F, n_abs, int(needed_from), dev)

        # long-run rows in time-major order (row u * A + a) through all layers: the spatial relations are time-major
        # anyway and the temporal edge list is simply numbered the same way, so nothing is ever transposed
        xl0 = torch.cat([cache['x_first'][:, :1], cache['x_main'][:, 1:]], dim=1).transpose(0, 1).reshape(NA, C)
        if D[0]:
            z0 = torch.cat([xl0, cache['x_first'][:, 1:].transpose(0, 1).reshape(-1, C)])     # special rows (f, d = 0, a)
        else:
            z0 = xl0
        t_src, t_dst, t_agent = cache['t_src'], cache['t_dst'], cache['t_agent']
        E_abs = t_src.numel()
        can_src, can_dst = t_src * A + t_agent, t_dst * A + t_agent
        # the long run's temporal relation is the same in every layer: ONE destination sort; a layer's special destinations
        # come after the long run's rows and their edge lists are produced already grouped by destination (below), so
        # they are appended to the CSR without sorting again
        rp_can, col_can, eid_can = ops.csr_build(torch.stack([can_src, can_dst]), NA)
        layers = []
        for l in range(1, NL + 1):
            Dp, Dl = D[l - 1], D[l]
            n_prev, n_own = (F - 1) * len(Dp) * A, (F - 1) * len(Dl) * A
            sh = shape['layers'][l - 1]
            rowptr, col, eid = rp_can, col_can, eid_can
            if n_own:
                # the special destinations' temporal edges: absolute edges into step f + d whose source lies inside frame f
                u_dst, f_of = sh['u_dst'], sh['f_of']
                hit = (t_dst.unsqueeze(0) == u_dst.unsqueeze(1)) & (t_src.unsqueeze(0) >= f_of.unsqueeze(1))
                j_sel, e_sel = hit.nonzero(as_tuple=True)
                ag, fj = t_agent[e_sel], f_of[j_sel]
                dp = t_src[e_sel] - fj                                                                  # source offset
                slot = sh['pos_in_prev'][dp.clamp(0, T)]
                is_special = dp <= (l - 1) * S
                src_special = NA + n_own + ((fj - 1) * max(len(Dp), 1) + slot.clamp_min(0)) * A + ag
                # (j_sel, e_sel) ascend in (j, edge) order and the cached edges are row-major in (agent, source step), so
                # the special destinations j * A + agent ascend with sources ascending inside a row: CSR order already
                first = torch.searchsorted(j_sel * A + ag, sh['own_rows']) + E_abs
                rowptr = torch.cat([rp_can[:NA], first.to(torch.int32),
                                    torch.full((n_prev,), E_abs + int(e_sel.numel()), dtype=torch.int32, device=dev)])
                col = torch.cat([col_can, torch.where(is_special, src_special, t_src[e_sel] * A + ag).to(torch.int32)])
                eid = torch.cat([eid_can, e_sel.to(torch.int32)])
            n_in = NA + n_own + (n_prev if n_own else 0)
            g_t = CSRGraph(None, n_in, n_in, rowptr, col, eid, num_feature_rows=int(cache['t_src'].numel()))

            # spatial relations over [long run | special rows (f, d, a)]
            if n_own:
                g_pl2a = cache['g_pl2a'].with_rows(sh['row_index'], None, P)
                g_a2a = cache['g_a2a'].with_rows(sh['row_index'], sh['src_offset'], NA + n_own)
            else:
                g_pl2a, g_a2a = cache['g_pl2a'], cache['g_a2a']
            layers.append(dict(n_own=n_own, own_idx=sh['own_idx'], g_t=g_t, g_pl2a=g_pl2a, g_a2a=g_a2a))
        return dict(z0=z0, layers=layers, index=shape['index'])

    _SHAPE_PLANS: 'collections.OrderedDict' = collections.OrderedDict()
    _SHAPE_PLAN_LIMIT = 8

    def _stream_shape_plan(self, D, A: int, F: int, n_abs: int, needed_from: int, dev) -> dict:
        """The part of :meth:`_stream_plan_graphs` that depends on SHAPES only (number of agents / frames / steps and the
        streaming plan ``D``): the gather of the special destinations' own rows, their (frame, offset) lists, the
        per-destination CSR-row / source-offset indirections of the spatial relations and the final (frame, agent, step)
        -> table-row index.  ~100 small launches of index arithmetic when built; kept per shape (a few MB each, the
        last ``_SHAPE_PLAN_LIMIT`` shapes) — batches of the same size reuse them, and the tensors are read-only."""
        T, S, NL = self.num_historical_steps, self.time_span, self.num_layers
        key = (T, S, NL, A, F, n_abs, needed_from, str(dev))
        plans = QCNetAgentEncoder._SHAPE_PLANS
        hit = plans.get(key)
        if hit is not None:
            plans.move_to_end(key)
            if hit['ready'] is not None:
                torch.cuda.current_stream(dev).wait_event(hit['ready'])      # built on whichever stream was current then
            return hit
        NA = A * n_abs
        i64 = dict(device=dev, dtype=torch.int64)
        a_ids = torch.arange(A, **i64)
        frames = torch.arange(1, F, **i64)                                            # frames with special steps
        layers = []
        for l in range(1, NL + 1):
            Dp, Dl = D[l - 1], D[l]
            n_own = (F - 1) * len(Dl) * A
            sh = dict(own_idx=None)
            if n_own:
                # the destinations' own rows (residual / query input): previous-layer value at (f, d)
                own = []
                for d in Dl:
                    if d in Dp:
                        own.append(NA + (((frames - 1) * len(Dp) + Dp.index(d)) * A).unsqueeze(1) + a_ids.unsqueeze(0))
                    else:
                        own.append(((frames + d) * A).unsqueeze(1) + a_ids.unsqueeze(0))                # [F-1, A]
                sh['own_idx'] = torch.stack(own, dim=1).reshape(n_own)
                d_vec = torch.tensor(Dl, **i64)
                sh['u_dst'] = (frames.unsqueeze(1) + d_vec.unsqueeze(0)).reshape(-1)                    # [J]
                sh['f_of'] = frames.unsqueeze(1).expand(-1, len(Dl)).reshape(-1).contiguous()
                pos_in_prev = torch.full((T + 1,), -1, **i64)
                if Dp:
                    pos_in_prev[torch.tensor(Dp, **i64)] = torch.arange(len(Dp), **i64)
                sh['pos_in_prev'] = pos_in_prev
                sh['own_rows'] = torch.arange(n_own + 1, **i64)
                u_rows = (sh['u_dst'].reshape(-1, 1) * A + a_ids.unsqueeze(0)).reshape(-1)
                sh['row_index'] = torch.cat([torch.arange(NA, **i64), u_rows]).to(torch.int32)
                blk = NA + torch.arange(n_own, **i64)
                sh['src_offset'] = torch.cat([torch.zeros(NA, **i64), blk - blk % A - (u_rows - u_rows % A)]).to(torch.int32)
            layers.append(sh)
        # every needed (frame, agent, step) as a row of ONE table: the long run's rows (u, a) followed by the special rows
        # of the last layer — consecutive frames share all but one of their steps, so the table has
        # n_abs + (F - 1) |D| rows per agent (time-major) where the dense [F, A, T] tensor has F * T
        tail = torch.arange(needed_from, T, **i64)
        steps = torch.arange(F, **i64).unsqueeze(1) + tail.unsqueeze(0)                                   # [F, Mt]
        index = (steps.view(F, 1, -1) * A + a_ids.view(1, A, 1)).contiguous()                            # [F, A, Mt]
        for i, d in enumerate(D[NL]):
            index[1:, :, d - needed_from] = NA + ((frames - 1).view(-1, 1) * len(D[NL]) + i) * A + a_ids.view(1, A)
        plan = dict(layers=layers, index=index, ready=None)
        if dev.type == 'cuda':
            plan['ready'] = torch.cuda.current_stream(dev).record_event()
        plans[key] = plan
        while len(plans) > QCNetAgentEncoder._SHAPE_PLAN_LIMIT:
            plans.popitem(last=False)
        return plan

    @torch.no_grad()
    def forward_frames_streaming(self, data, map_enc: Mapping[str, torch.Tensor], cache, needed_from: int) -> torch.Tensor:
        """Dense form of :meth:`stream_frames`: ``x_a`` [F * A, T, C] like :meth:`forward_frames`, entries before
        ``needed_from`` left uninitialised."""
        T, C = self.num_historical_steps, self.hidden_dim
        table, index = self.stream_frames(data, map_enc, cache, needed_from)
        F, A = index.size(0), index.size(1)
        out = torch.empty(F, A, T, C, dtype=table.dtype, device=table.device)
        out[:, :, int(needed_from):] = table[index]
        return out.view(F * A, T, C)

    def forward(self, data, map_enc: Mapping[str, torch.Tensor], time_shift: int, cache=None,
                return_edge_index: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """``cache``: result of :meth:`precompute` for a range of time shifts containing ``time_shift`` (same output,
        the shift-invariant work is not repeated).  The ``edge_index_*`` entries are this implementation's extras
        (the reference returns only ``x_a``); with a cache they are produced only on request."""
        if cache is not None:
            return self._forward_cached(data, map_enc, time_shift, cache, bool(return_edge_index))
        T, ts = self.num_historical_steps, int(time_shift)
        agent = data['agent']
        A = int(agent['num_nodes'])
        mask = agent['valid_mask'][:, ts:T + ts].contiguous()
        pos_a = agent['position'][:, ts:T + ts, :2].contiguous()
        head_a = agent['heading'][:, ts:T + ts].contiguous()
        vel = agent['velocity'][:, ts:T + ts, :2].contiguous()
        prev = torch.cat([pos_a[:, :1], pos_a[:, :-1]], dim=1)           # zero motion vector at the first step (:99)
        # one type embedding per agent, shared by its T steps (reference: repeat_interleave(T), :109)
        x_a = self.x_a_emb(continuous_inputs=self._agent_features(pos_a, prev, head_a, vel).view(-1, 4),
                           categorical_embs=[self.type_a_emb(agent['type'].long())], categorical_repeat=T)

        pos_t, head_t = pos_a.reshape(-1, 2), head_a.reshape(-1)
        ei_t = self._temporal_edges(mask)
        g_t = CSRGraph.from_edge_index(ei_t, A * T, A * T, validate=False)
        r_t = NormalizedEdgeFeatures(self.r_t_emb(
            continuous_inputs=rel_geometry(pos_t, pos_t, head_t, head_t, ei_t, extra=ei_t[0] - ei_t[1]),
            categorical_embs=None, normalized_output=True))
        rel = self._spatial_relations(data, pos_a, head_a, mask, T)
        x_a = self._layers(x_a.view(-1, T, self.hidden_dim), map_enc['x_pl'][:, 0], g_t, r_t, rel['g_pl2a'],
                           rel['r_pl2a'], rel['g_a2a'], rel['r_a2a'])
        return {'x_a': x_a, 'edge_index_t': ei_t, 'edge_index_pl2a': rel['g_pl2a'].edge_index,
                'edge_index_a2a': rel['g_a2a'].edge_index}
