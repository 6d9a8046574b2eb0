"""QCNetMapEncoder — drop-in for modules/qcnet_map_encoder.py:30-173 (same ctor kwargs, forward(data) and
state_dict keys; ``input_dim == 2`` / ``argoverse_v2`` branch).

Differences that do not change results: the polygon-to-polygon radius graph comes from the HIP radius kernel;
both relations are turned into destination-sorted CSR once; ``x_pl`` is returned as a stride-0 [P, T, C] view
of the [P, C] encoding instead of T materialised copies (reference :170-171).
"""
from typing import Dict

import torch
import torch.nn as nn

from .. import ops
from ..layers import attention_layer
from ..layers import AttentionLayer, FourierEmbedding, NormalizedEdgeFeatures
from ..utils.graph import CSRGraph, merge_edges
from ..utils.weight_init import weight_init
from ..utils.prep import run_prep
from ._common import node_ptr, rel_geometry


class QCNetMapEncoder(nn.Module):

    def __init__(self,
                 dataset: str,
                 input_dim: int,
                 hidden_dim: int,
                 num_historical_steps: int,
                 pl2pl_radius: float,
                 num_freq_bands: int,
                 num_layers: int,
                 num_heads: int,
                 head_dim: int,
                 dropout: float) -> None:
        super().__init__()
        if dataset != 'argoverse_v2':
            raise ValueError('{} is not a valid dataset'.format(dataset))
        if input_dim not in (2, 3):
            raise ValueError('{} is not a valid dimension'.format(input_dim))
        self.dataset = dataset
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_historical_steps = num_historical_steps
        self.pl2pl_radius = pl2pl_radius
        self.num_freq_bands = num_freq_bands
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.dropout = dropout

        for name, vocab in (('type_pt_emb', 17), ('side_pt_emb', 3), ('type_pl_emb', 4), ('int_pl_emb', 3),
                            ('type_pl2pl_emb', 5)):
            setattr(self, name, nn.Embedding(vocab, hidden_dim))
        dims = (1, 0, 3, 3) if input_dim == 2 else (2, 1, 4, 4)       # reference :56-65: 3-D adds heights and a dz column
        for name, dim in zip(('x_pt_emb', 'x_pl_emb', 'r_pt2pl_emb', 'r_pl2pl_emb'), dims):
            setattr(self, name, FourierEmbedding(input_dim=dim, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands))
        layer = dict(hidden_dim=hidden_dim, num_heads=num_heads, head_dim=head_dim, dropout=dropout, has_pos_emb=True)
        self.pt2pl_layers = nn.ModuleList([AttentionLayer(bipartite=True, **layer) for _ in range(num_layers)])
        self.pl2pl_layers = nn.ModuleList([AttentionLayer(bipartite=False, **layer) for _ in range(num_layers)])
        self.apply(weight_init)

    def forward(self, data) -> Dict[str, torch.Tensor]:
        def prepare():
            # everything that depends on the map alone: input embeddings, both relations with their positional
            # embeddings (runs on the preparation stream, utils/prep.py)
            pt, pl = data['map_point'], data['map_polygon']
            pos_pt = pt['position'][:, :2].contiguous()
            orient_pt = pt['orientation'].contiguous()
            pos_pl = pl['position'][:, :2].contiguous()
            orient_pl = pl['orientation'].contiguous()
            n_pt, n_pl = pos_pt.size(0), pos_pl.size(0)

            dim3 = self.input_dim == 3
            if dim3:   # reference :106-108
                x_pt_in = torch.stack([pt['magnitude'], pt['height']], dim=-1)
                x_pl_in = pl['height'].unsqueeze(-1)
                z_pt, z_pl = pt['position'][:, 2], pl['position'][:, 2]
            else:
                x_pt_in, x_pl_in = pt['magnitude'].unsqueeze(-1), None

            def geometry(pos_s, pos_d, orient_s, orient_d, ei, z_s=None):
                """[dist, angle, d_orient] rows (one kernel); 3-D: the height difference as third column (:129-135,156-162)"""
                r = rel_geometry(pos_s, pos_d, orient_s, orient_d, ei)
                if not dim3:
                    return r
                return torch.stack([r[:, 0], r[:, 1], z_s[ei[0]] - z_pl[ei[1]], r[:, 2]], dim=-1).contiguous()

            x_pt = self.x_pt_emb(continuous_inputs=x_pt_in,
                                 categorical_embs=[self.type_pt_emb(pt['type'].long()), self.side_pt_emb(pt['side'].long())])
            x_pl = self.x_pl_emb(continuous_inputs=x_pl_in,
                                 categorical_embs=[self.type_pl_emb(pl['type'].long()),
                                                   self.int_pl_emb(pl['is_intersection'].long())])

            ei_pt2pl = data['map_point', 'to', 'map_polygon']['edge_index']
            g_pt2pl = CSRGraph.from_edge_index(ei_pt2pl, n_pt, n_pl)
            # the embedding launches themselves go to the main stream (utils/prep.py::defer)
            geo_pt2pl = geometry(pos_pt, pos_pl, orient_pt, orient_pl, ei_pt2pl, z_pt if dim3 else None)
            r_pt2pl = NormalizedEdgeFeatures.deferred(lambda: self.r_pt2pl_emb(
                continuous_inputs=geo_pt2pl, categorical_embs=None, normalized_output=True), (geo_pt2pl,))

            ptr_pl = node_ptr(data, 'map_polygon', pos_pl.device)
            ei_radius, _ = ops.radius_pairs(pos_pl, pos_pl, self.pl2pl_radius, ptr_pl, ptr_pl, 300 + 1, drop_self=True,
                                            y_first=False)
            type_data = data['map_polygon', 'to', 'map_polygon']['type']
            ei_pl2pl, type_pl2pl = merge_edges(
                edge_indices=[ei_radius, data['map_polygon', 'to', 'map_polygon']['edge_index']],
                edge_attrs=[type_data.new_zeros(ei_radius.size(1), dtype=torch.uint8), type_data], reduce='max')
            g_pl2pl = CSRGraph.from_edge_index(ei_pl2pl, n_pl, n_pl)
            geo_pl2pl = geometry(pos_pl, pos_pl, orient_pl, orient_pl, ei_pl2pl, z_pl if dim3 else None)
            r_pl2pl = NormalizedEdgeFeatures.deferred(lambda: self.r_pl2pl_emb(
                continuous_inputs=geo_pl2pl, categorical_embs=[self.type_pl2pl_emb(type_pl2pl.long())],
                normalized_output=True), (geo_pl2pl, type_pl2pl))

            return x_pt, x_pl, g_pt2pl, r_pt2pl, g_pl2pl, r_pl2pl, ei_pl2pl, type_pl2pl

        x_pt, x_pl, g_pt2pl, r_pt2pl, g_pl2pl, r_pl2pl, ei_pl2pl, type_pl2pl = run_prep(
            data['map_polygon']['position'].device, prepare)
        # chained stack (layers/attention_layer.py: each layer's node-side kernel also projects the next layer's destinations)
        stack = []
        for i in range(self.num_layers):
            stack += [(self.pt2pl_layers[i], x_pt, r_pt2pl, g_pt2pl), (self.pl2pl_layers[i], None, r_pl2pl, g_pl2pl)]
        pre = None
        for j, (layer, x_src, r, g) in enumerate(stack):
            nxt = stack[j + 1][0] if j + 1 < len(stack) and attention_layer.CHAIN_NODE_SIDE else None
            res = layer(x_pl if x_src is None else (x_src, x_pl), r, g, pre=pre, next_layer=nxt)
            x_pl, pre = res if nxt is not None else (res, None)
        x_pl = x_pl.unsqueeze(1).expand(-1, self.num_historical_steps, -1)
        return {'x_pt': x_pt, 'x_pl': x_pl, 'edge_index_pl2pl': ei_pl2pl, 'type_pl2pl': type_pl2pl}
