"""QCNetAgentEncoder — drop-in for modules/qcnet_agent_encoder.py:32-189 (same ctor kwargs,
forward(data, map_enc, time_shift) and state_dict keys).

Graphs: the temporal graph keeps the reference's dense_to_sparse order; polygon→agent and agent→agent edges come
from the HIP radius kernel with the validity filters of reference :160,170 fused in; every relation is converted
to a destination-sorted CSR once and reused by both layers.  The T-fold tiling of the polygon features
(reference :184) is folded away: sources are gathered from the single [P, C] copy.
"""
from typing import Dict, Mapping, Optional

import torch
import torch.nn as nn

from .. import ops
from ..layers import AttentionLayer, FourierEmbedding, NormalizedEdgeFeatures
from ..utils.geometry import angle_between_2d_vectors
from ..utils.graph import CSRGraph, dense_to_sparse
from ..utils.weight_init import weight_init
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

    def forward(self, data, map_enc: Mapping[str, torch.Tensor], time_shift: int) -> Dict[str, torch.Tensor]:
        T, C, ts = self.num_historical_steps, self.hidden_dim, int(time_shift)
        agent = data['agent']
        A = int(agent['num_nodes'])
        P = int(data['map_polygon']['num_nodes'])
        mask = agent['valid_mask'][:, ts:T + ts].contiguous()
        pos_a = agent['position'][:, ts:T + ts, :2].contiguous()
        dev = pos_a.device
        motion = torch.cat([pos_a.new_zeros(A, 1, 2), pos_a[:, 1:] - pos_a[:, :-1]], dim=1)
        head_a = agent['heading'][:, ts:T + ts].contiguous()
        head_vec_a = torch.stack([head_a.cos(), head_a.sin()], dim=-1)
        pos_pl = data['map_polygon']['position'][:, :2].contiguous()
        orient_pl = data['map_polygon']['orientation'].contiguous()
        vel = agent['velocity'][:, ts:T + ts, :2].contiguous()

        x_a = torch.stack([torch.norm(motion, p=2, dim=-1),
                           angle_between_2d_vectors(ctr_vector=head_vec_a, nbr_vector=motion),
                           torch.norm(vel, p=2, dim=-1),
                           angle_between_2d_vectors(ctr_vector=head_vec_a, nbr_vector=vel)], dim=-1)
        # one type embedding per agent, shared by its T steps (reference: repeat_interleave(T), :109)
        x_a = self.x_a_emb(continuous_inputs=x_a.view(-1, 4), categorical_embs=[self.type_a_emb(agent['type'].long())],
                           categorical_repeat=T)

        # temporal relation, agent-major node id a * T + t: dst later than src by at most time_span steps
        pos_t, head_t = pos_a.reshape(-1, 2), head_a.reshape(-1)
        ei_t = dense_to_sparse(mask.unsqueeze(2) & mask.unsqueeze(1))[0]
        ei_t = ei_t[:, ei_t[1] > ei_t[0]]
        ei_t = ei_t[:, ei_t[1] - ei_t[0] <= self.time_span]
        g_t = CSRGraph.from_edge_index(ei_t, A * T, A * T, validate=False)
        r_t = NormalizedEdgeFeatures(self.r_t_emb(
            continuous_inputs=rel_geometry(pos_t, pos_t, head_t, head_t, ei_t, extra=ei_t[0] - ei_t[1]),
            categorical_embs=None, normalized_output=True))

        # spatial relations, time-major node id t * A + a; one radius example per (time step, scenario)
        pos_s = pos_a.transpose(0, 1).reshape(-1, 2).contiguous()
        head_s = head_a.transpose(0, 1).reshape(-1)
        mask_s = mask.transpose(0, 1).reshape(-1).contiguous()
        pos_pl_rep = pos_pl.repeat(T, 1)
        orient_pl_rep = orient_pl.repeat(T)
        steps = torch.arange(T, device=dev, dtype=torch.int64).unsqueeze(1)
        ptr_a, ptr_p = node_ptr(data, 'agent', dev), node_ptr(data, 'map_polygon', dev)
        ptr_s = torch.cat([(steps * A + ptr_a[:-1]).reshape(-1), ptr_a.new_tensor([T * A])])
        ptr_pl = torch.cat([(steps * P + ptr_p[:-1]).reshape(-1), ptr_p.new_tensor([T * P])])

        ei_pl2a, _ = ops.radius_pairs(pos_s, pos_pl_rep, self.pl2a_radius, ptr_s, ptr_pl, 300, mask_x=mask_s)
        g_pl2a = CSRGraph.from_edge_index(ei_pl2a, T * P, T * A, validate=False).fold_sources(P)
        r_pl2a = NormalizedEdgeFeatures(self.r_pl2a_emb(
            continuous_inputs=rel_geometry(pos_pl_rep, pos_s, orient_pl_rep, head_s, ei_pl2a), categorical_embs=None, normalized_output=True))

        ei_a2a, off_a2a = ops.radius_pairs(pos_s, pos_s, self.a2a_radius, ptr_s, ptr_s, 300 + 1, drop_self=True,
                                           mask_x=mask_s, mask_y=mask_s, y_first=False)
        g_a2a = CSRGraph.from_sorted(ei_a2a, off_a2a, T * A, T * A)
        r_a2a = NormalizedEdgeFeatures(self.r_a2a_emb(
            continuous_inputs=rel_geometry(pos_s, pos_s, head_s, head_s, ei_a2a), categorical_embs=None, normalized_output=True))

        x_pl = map_enc['x_pl'][:, 0]  # [P, C]; the T copies of reference :170 are identical
        for i in range(self.num_layers):
            x_a = x_a.reshape(-1, C)
            x_a = self.t_attn_layers[i](x_a, r_t, g_t)
            x_a = x_a.reshape(-1, T, C).transpose(0, 1).reshape(-1, C)
            x_a = self.pl2a_attn_layers[i]((x_pl, x_a), r_pl2a, g_pl2a)
            x_a = self.a2a_attn_layers[i](x_a, r_a2a, g_a2a)
            x_a = x_a.reshape(T, -1, C).transpose(0, 1)
        return {'x_a': x_a, 'edge_index_t': ei_t, 'edge_index_pl2a': ei_pl2a, 'edge_index_a2a': ei_a2a}
