"""QCNetDecoder — drop-in for modules/qcnet_decoder.py:35-294 (same ctor kwargs,
forward(data, scene_enc, time_shift), output dict and state_dict keys; ``output_head=False`` branch).

The propose/refine stacks run on destination-sorted CSR graphs built once per call.  The K-fold tiling of the
polygon and agent features (reference :150-151) is folded away: sources are gathered from the single copy.
``TemporalAggregate`` shares this implementation through the ``_fuse_modes`` hook.
"""
import math
import os
from typing import Dict, List, Mapping, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..layers import attention_layer
from ..layers import AttentionLayer, FourierEmbedding, MLPLayer, NormalizedEdgeFeatures
from ..utils.geometry import wrap_angle
from ..utils.graph import CSRGraph, mask_outer_edges
from ..utils.weight_init import weight_init
from ..utils.prep import run_prep
from ._common import node_ptr, rel_geometry

# the GRU's input halves in the recurrence's own blocked layout (ops.trajectory_gru); QCN_BLOCKED_GRU=0: row-major, for A/B runs
BLOCKED_GRU_INPUT = os.environ.get('QCN_BLOCKED_GRU', '1') != '0'


class QCNetDecoder(nn.Module):

    def __init__(self,
                 dataset: str,
                 input_dim: int,
                 hidden_dim: int,
                 output_dim: int,
                 output_head: bool,
                 num_historical_steps: int,
                 num_future_steps: int,
                 num_modes: int,
                 num_recurrent_steps: int,
                 num_t2m_steps: Optional[int],
                 pl2m_radius: float,
                 a2m_radius: float,
                 num_freq_bands: int,
                 num_layers: int,
                 num_heads: int,
                 head_dim: int,
                 dropout: float) -> None:
        super().__init__()
        if output_head and num_future_steps % num_recurrent_steps != 0:
            raise ValueError('num_future_steps must be divisible by num_recurrent_steps')
        if (num_future_steps * output_dim) % num_recurrent_steps != 0:
            raise ValueError('num_future_steps * output_dim must be divisible by num_recurrent_steps')
        self.dataset = dataset
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim
        self.output_head = output_head
        self.num_historical_steps = num_historical_steps
        self.num_future_steps = num_future_steps
        self.num_modes = num_modes
        self.num_recurrent_steps = num_recurrent_steps
        self.num_t2m_steps = num_t2m_steps if num_t2m_steps is not None else num_historical_steps
        self.pl2m_radius = pl2m_radius
        self.a2m_radius = a2m_radius
        self.num_freq_bands = num_freq_bands
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.dropout = dropout

        self.mode_emb = nn.Embedding(num_modes, hidden_dim)
        for name, dim in (('r_t2m_emb', 4), ('r_pl2m_emb', 3), ('r_a2m_emb', 3), ('y_emb', output_dim + output_head)):
            setattr(self, name, FourierEmbedding(input_dim=dim, hidden_dim=hidden_dim, num_freq_bands=num_freq_bands))
        self.traj_emb = nn.GRU(input_size=hidden_dim, hidden_size=hidden_dim, num_layers=1, bias=True,
                               batch_first=False, dropout=0.0, bidirectional=False)
        self.traj_emb_h0 = nn.Parameter(torch.zeros(1, hidden_dim))
        layer = dict(hidden_dim=hidden_dim, num_heads=num_heads, head_dim=head_dim, dropout=dropout)
        for stage in ('propose', 'refine'):
            for rel in ('t2m', 'pl2m', 'a2m'):
                setattr(self, '%s_%s_attn_layers' % (rel, stage), nn.ModuleList(
                    [AttentionLayer(bipartite=True, has_pos_emb=True, **layer) for _ in range(num_layers)]))
            setattr(self, 'm2m_%s_attn_layer' % stage, AttentionLayer(bipartite=False, has_pos_emb=False, **layer))
        step_dim = num_future_steps * output_dim // num_recurrent_steps
        self.to_loc_propose_pos = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=step_dim)
        self.to_scale_propose_pos = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=step_dim)
        self.to_loc_refine_pos = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim,
                                          output_dim=num_future_steps * output_dim)
        self.to_scale_refine_pos = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim,
                                            output_dim=num_future_steps * output_dim)
        if output_head:   # heading outputs (reference :125-132)
            head_step = num_future_steps // num_recurrent_steps
            self.to_loc_propose_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=head_step)
            self.to_conc_propose_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=head_step)
            self.to_loc_refine_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=num_future_steps)
            self.to_conc_refine_head = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=num_future_steps)
        else:
            self.to_loc_propose_head = None
            self.to_conc_propose_head = None
            self.to_loc_refine_head = None
            self.to_conc_refine_head = None
        self.to_pi = MLPLayer(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=1)
        self.apply(weight_init)

    # the tensors forward() reads of its ``data`` argument (data.frame_batch replicates only these for the ten-frame pass)
    DATA_FIELDS = {'agent': ('position', 'heading', 'valid_mask', 'predict_mask'),
                   'map_polygon': ('position', 'orientation')}

    def _gru_packs(self):
        """traj_emb parameters in the kernels' layouts (k-major transposes), cached per parameter version."""
        gru = self.traj_emb
        out_lin = self.y_emb.to_out[2]
        prm = (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0, self.traj_emb_h0, out_lin.weight, out_lin.bias)
        mode = ops.matmul_mode()
        key = (mode,) + tuple((p.data_ptr(), p._version) for p in prm)
        if key != getattr(self, '_gru_key', None):
            C = self.hidden_dim
            with torch.no_grad():
                # y_emb's last Linear feeds the GRU's input half directly (:255-257): folded into it, one GEMM per row less.
                # Not in the bf16 mode: there every Linear of the reference multiplies its own bf16-rounded operands (the
                # mode's contract, include/qcnet_hip.h), and a folded product would round differently.
                fold = mode != ops.MATMUL_BF16
                if fold:
                    w_f, b_f = ops.fold_linear(list(gru.weight_ih_l0.view(3, C, C)), list(gru.bias_ih_l0.view(3, C)),
                                               out_lin.weight, out_lin.bias)
                else:
                    w_f, b_f = list(gru.weight_ih_l0.view(3, C, C)), gru.bias_ih_l0.view(3, C).contiguous()
                wih_t = ops.pack_chain_matrices(w_f, mode=mode)
                self._gru_pack = (wih_t, b_f, ops.pack_gru_whh(gru.weight_hh_l0, self.traj_emb_h0, mode=mode),
                                  gru.bias_hh_l0.contiguous(), self.traj_emb_h0.reshape(-1).contiguous(), mode, fold)
            self._gru_key = key
        return self._gru_pack

    def _head_cache(self, name: str) -> dict:
        caches = self.__dict__.setdefault('_mlp_head_caches', {})
        return caches.setdefault(name, {})

    # hook: TemporalAggregate sums the stored mode queries of the earlier frames into the new ones here
    def _fuse_modes(self, m: torch.Tensor, temporal_m: Optional[torch.Tensor]) -> torch.Tensor:
        return m

    _mode_key = 'm'

    def _cross_stack(self, stage: str, m: torch.Tensor, x_t, x_pl, x_a, r_t2m, r_pl2m, r_a2m,
                     g_t2m, g_pl2m, g_a2m, g_m2m, source_cache: Optional[dict] = None) -> torch.Tensor:
        K, C = self.num_modes, self.hidden_dim
        t2m = getattr(self, 't2m_%s_attn_layers' % stage)
        pl2m = getattr(self, 'pl2m_%s_attn_layers' % stage)
        a2m = getattr(self, 'a2m_%s_attn_layers' % stage)
        # The reference flips the mode queries between agent-major (t2m, m2m) and mode-major (pl2m, a2m) node order
        # (:223,226).  The kernel takes the node order of a mode-replicated relation as a parameter, so m stays
        # agent-major throughout and the four transposes per layer pair disappear.
        m = m.reshape(-1, C)
        # every layer's output is the next layer's destination input and nothing else: the stack runs chained (each layer's
        # node-side kernel also applies the next layer's destination-side LayerNorm + projections, layers/attention_layer.py)
        stack = []
        for i in range(self.num_layers):
            stack += [(t2m[i], x_t, r_t2m, g_t2m), (pl2m[i], x_pl, r_pl2m, g_pl2m), (a2m[i], x_a, r_a2m, g_a2m)]
        stack.append((getattr(self, 'm2m_%s_attn_layer' % stage), None, None, g_m2m))
        pre = None
        for j, (layer, x_src, r, g) in enumerate(stack):
            nxt = stack[j + 1][0] if j + 1 < len(stack) and attention_layer.CHAIN_NODE_SIDE else None
            res = layer(m if x_src is None else (x_src, m), r, g, source_cache if x_src is not None else None,
                        pre=pre, next_layer=nxt)
            m, pre = res if nxt is not None else (res, None)
        return m.reshape(-1, K, C)

    def _decode(self, data, scene_enc: Mapping[str, torch.Tensor], time_shift: int,
                temporal_m: Optional[torch.Tensor], return_edge_index: bool = True) -> Dict[str, torch.Tensor]:
        return self._refine_phase(self._propose_phase(data, scene_enc, time_shift, return_edge_index), temporal_m)

    def _propose_phase(self, data, scene_enc: Mapping[str, torch.Tensor], time_shift: int,
                       return_edge_index: bool = True) -> dict:
        """Everything up to and including the GRU re-encoding of the proposals (reference :141-257): graphs, the recurrent
        proposal steps, the trajectory embedding.  Independent of ``temporal_m`` — TemporalAggregate's copy of this phase
        can therefore run beside the ten-frame decoder pass whose mode queries it later sums in (see
        predictors/temp_ens_learn_agg.py)."""
        T, C, K, ts = self.num_historical_steps, self.hidden_dim, self.num_modes, int(time_shift)
        Tf, OD = self.num_future_steps, self.output_dim
        agent = data['agent']
        A = int(agent['num_nodes'])
        P = int(data['map_polygon']['num_nodes'])
        dev = agent['position'].device

        # only the last num_t2m_steps steps are ever sources of the time->mode attention (:143): project those alone.
        # A streamed encoding (QCNetAgentEncoder.stream_frames) comes as a table of distinct rows + an index per
        # (agent, step): sliding frames share their steps, so the table is several times smaller than [A, Mt].
        Mt = min(self.num_t2m_steps, T)
        t_index = None
        if 'x_t_table' in scene_enc:
            x_t, t_index = scene_enc['x_t_table'], scene_enc['x_t_index']                  # [N, C], [A, Mt] int64
            if tuple(t_index.shape) != (A, Mt):
                raise ValueError('x_t_index must be [%d, %d]' % (A, Mt))
            x_a_now = x_t[t_index[:, -1]]
        else:
            x_t = scene_enc['x_a'][:, T - Mt:].reshape(-1, C)
            x_a_now = scene_enc['x_a'][:, -1]
        x_pl = scene_enc['x_pl'][:, T - 1]          # [P, C]; reference tiles it K times (:150)
        # a frame batch replicates the map per frame: the encoding may come as ONE copy ([P / F, C]), sources are folded
        n_pl_rows = x_pl.size(0)
        if n_pl_rows != P and (n_pl_rows == 0 or P % n_pl_rows != 0):
            raise ValueError('x_pl has %d rows for %d polygons' % (n_pl_rows, P))
        x_a = x_a_now                               # [A, C]; reference tiles it K times (:151)
        m = self.mode_emb.weight.repeat(A, 1)
        n_t_rows = x_t.size(0)

        def prepare():
            """Masks, edge lists, CSR graphs and positional embeddings of the four relations: functions of ``data`` (and
            of the index half of a streamed encoding) only — run on the preparation stream (utils/prep.py)."""
            pos_m = agent['position'][:, T - 1 + ts, :2].contiguous()
            head_m = agent['heading'][:, T - 1 + ts].contiguous()
            # the three mode relations are consumed by qcn_attn_modes_f32 only: embeddings straight in its operand format
            r_fmt = True
            if attention_layer.USE_ATTN_MODES and 1 < K <= 6 and os.environ.get('QCN_R_SPLIT_FUSED', '1') != '0':
                r_fmt = 'bf16' if ops.matmul_mode() == ops.MATMUL_BF16 else 'split'
            mask_src = agent['valid_mask'][:, ts:T + ts].clone()
            mask_src[:, :T - self.num_t2m_steps] = False
            mask_dst = agent['predict_mask'].any(dim=-1, keepdim=True).repeat(1, K)

            pos_t = agent['position'][:, ts:T + ts, :2].reshape(-1, 2)
            head_t = agent['heading'][:, ts:T + ts].reshape(-1)
            ei_t2m_1, off_t2m = ops.mask_pairs(mask_src, mask_dst[:, -1:])
            geo_t2m = rel_geometry(pos_t, pos_m, head_t, head_m, ei_t2m_1, extra=(ei_t2m_1[0] % T) - T + 1)
            # One positional embedding / edge list per (step, agent) pair; the K mode queries of the agent share it
            # (reference: r_t2m.repeat_interleave(K) and a K-times larger edge list, :169-170).  The embedding launches
            # themselves go to the main stream (utils/prep.py::defer).
            r_t2m = NormalizedEdgeFeatures.deferred(lambda: self.r_t2m_emb(
                continuous_inputs=geo_t2m, categorical_embs=None, normalized_output=r_fmt), (geo_t2m,), rep=K)
            # the reference's K-fold replicated COO lists (:169-170,190-191,208-210,213) are outputs for inspection / parity
            # checks only — the kernels work on the base lists — and are built on request (one enumeration launch with a
            # host round trip and two K-fold concatenations per call otherwise)
            ei_t2m = mask_outer_edges(mask_src, mask_dst) if return_edge_index else None
            # the enumeration is row-major over (agent, step): already grouped by destination agent, sources ascending —
            # the CSR needs no sort, its row pointer is every T-th entry of the enumeration's per-source-row offsets
            col = ei_t2m_1[0]                           # a * T + t  ->  a * Mt + (t - (T - Mt))
            col = col - (torch.div(col, T, rounding_mode='floor') + 1) * (T - Mt)
            if t_index is not None:
                col = t_index.reshape(-1)[col]
            g_t2m = CSRGraph(ei_t2m_1, n_t_rows, A, off_t2m[::T].to(torch.int32), col.to(torch.int32), None)
            g_t2m = g_t2m.replicate(K, ops.LAYOUT_AGENT_MAJOR, edge_index=ei_t2m)

            pos_pl = data['map_polygon']['position'][:, :2].contiguous()
            orient_pl = data['map_polygon']['orientation'].contiguous()
            ptr_a, ptr_p = node_ptr(data, 'agent', dev), node_ptr(data, 'map_polygon', dev)
            dst_ok = mask_dst[:, 0].contiguous()
            ei_pl2m_1, _ = ops.radius_pairs(pos_m, pos_pl, self.pl2m_radius, ptr_a, ptr_p, 300, mask_x=dst_ok)
            geo_pl2m = rel_geometry(pos_pl, pos_m, orient_pl, head_m, ei_pl2m_1)
            r_pl2m = NormalizedEdgeFeatures.deferred(lambda: self.r_pl2m_emb(
                continuous_inputs=geo_pl2m, categorical_embs=None, normalized_output=r_fmt), (geo_pl2m,), rep=K)
            ei_pl2m = None
            if return_edge_index:
                ei_pl2m = torch.cat([ei_pl2m_1 + i * ei_pl2m_1.new_tensor([[P], [A]]) for i in range(K)], dim=1)
            g_pl2m = CSRGraph.from_edge_index(ei_pl2m_1, P, A, validate=False)
            if n_pl_rows != P:
                g_pl2m = g_pl2m.fold_sources(n_pl_rows)
            g_pl2m = g_pl2m.replicate(K, ops.LAYOUT_AGENT_MAJOR, edge_index=ei_pl2m)

            src_ok = mask_src[:, -1].contiguous()
            ei_a2m_1, off_a2m = ops.radius_pairs(pos_m, pos_m, self.a2m_radius, ptr_a, ptr_a, 300 + 1, drop_self=True,
                                                 mask_x=src_ok, mask_y=dst_ok, y_first=False)
            geo_a2m = rel_geometry(pos_m, pos_m, head_m, head_m, ei_a2m_1)
            r_a2m = NormalizedEdgeFeatures.deferred(lambda: self.r_a2m_emb(
                continuous_inputs=geo_a2m, categorical_embs=None, normalized_output=r_fmt), (geo_a2m,), rep=K)
            ei_a2m = torch.cat([ei_a2m_1 + i * A for i in range(K)], dim=1) if return_edge_index else None
            g_a2m = CSRGraph.from_sorted(ei_a2m_1, off_a2m, A, A).replicate(K, ops.LAYOUT_AGENT_MAJOR, edge_index=ei_a2m)

            # mode<->mode edges: a full K x K block per predicted agent, row-major — the relation is symmetric, so the
            # enumeration read with its rows swapped IS the destination-sorted list (sources ascending): no sort
            ei_m2m, off_m2m = ops.mask_pairs(mask_dst, mask_dst)
            g_m2m = CSRGraph(ei_m2m, A * K, A * K, off_m2m.to(torch.int32), ei_m2m[1].to(torch.int32), None)
            return (r_t2m, r_pl2m, r_a2m, g_t2m, g_pl2m, g_a2m, g_m2m, ei_pl2m_1, ei_t2m, ei_pl2m, ei_a2m, ei_m2m)

        (r_t2m, r_pl2m, r_a2m, g_t2m, g_pl2m, g_a2m, g_m2m,
         ei_pl2m_1, ei_t2m, ei_pl2m, ei_a2m, ei_m2m) = run_prep(dev, prepare)

        graphs = (x_t, x_pl, x_a, r_t2m, r_pl2m, r_a2m, g_t2m, g_pl2m, g_a2m, g_m2m)
        locs: List[torch.Tensor] = []
        scales: List[torch.Tensor] = []
        hlocs: List[torch.Tensor] = []
        hconcs: List[torch.Tensor] = []
        # the recurrent proposal steps apply the SAME layers to the SAME source rows: their key / value projections are
        # computed in the first step and kept (the reference recomputes them in every step, :219-227)
        source_cache: dict = {}
        for _ in range(self.num_recurrent_steps):
            m = self._cross_stack('propose', m, *graphs, source_cache=source_cache)
            loc, scale = MLPLayer.fused([self.to_loc_propose_pos, self.to_scale_propose_pos], m, self._head_cache('propose'))
            locs.append(loc)
            scales.append(scale)
            if self.output_head:
                hloc, hconc = MLPLayer.fused([self.to_loc_propose_head, self.to_conc_propose_head], m,
                                             self._head_cache('propose_head'))
                hlocs.append(hloc)
                hconcs.append(hconc)
        # cumulative way-points / scales over the future steps (reference :234-250): one launch per output
        # (qcn_proposal_cumsum_f32) instead of cat + outer-dim scan + elementwise ops
        flat = lambda chunks: [c.reshape(-1, c.size(-1)) for c in chunks]
        locs, scales, hlocs, hconcs = flat(locs), flat(scales), flat(hlocs), flat(hconcs)
        loc_propose_pos = ops.proposal_cumsum(locs, OD).view(-1, K, Tf, OD)
        scale_propose_pos = ops.proposal_cumsum(scales, OD, ops.CUMSUM_ELU1, 0.1).view(-1, K, Tf, OD)
        if self.output_head:   # reference :243-250: cumulative headings, concentrations; y_emb sees [x, y, wrapped heading]
            loc_propose_head = ops.proposal_cumsum(hlocs, 1, ops.CUMSUM_TANH_PI).view(-1, K, Tf, 1)
            conc_propose_head = ops.proposal_cumsum(hconcs, 1, ops.CUMSUM_ELU1, 0.02, reciprocal=True).view(-1, K, Tf, 1)
            y_in = torch.cat([loc_propose_pos.detach(), wrap_angle(loc_propose_head.detach())], dim=-1)
        else:
            loc_propose_head = loc_propose_pos.new_zeros((A, K, Tf, 1))
            conc_propose_head = scale_propose_pos.new_zeros((A, K, Tf, 1))
            y_in = loc_propose_pos.detach()
        y_dim = OD + int(self.output_head)

        # trajectory re-encoding: Fourier embedding of the proposed way-points, GRU over the Tf steps, last hidden state
        # (reference :255-257); row n * Tf + t of the embedding is step t of mode query n
        wih_t, b_ih, whh, b_hh, h0, mm_mode, fold = self._gru_packs()
        y_rows = y_in.reshape(-1, y_dim).contiguous().float()
        if mm_mode != ops.MATMUL_EXACT_F32 and BLOCKED_GRU_INPUT:
            # y_emb + the GRU's input half + the recurrence, the input halves in the recurrence's own layout (two launches)
            m = ops.trajectory_gru(y_rows, y_dim, Tf, self.y_emb.packed_weights(), wih_t, b_ih, fold, whh, b_hh, h0, mode=mm_mode)
        else:
            gi = ops.fourier_linear(y_rows, y_dim, self.y_emb.packed_weights(), wih_t, b_ih, folded=fold, mode=mm_mode)
            m = ops.gru_recurrence(gi[0], gi[1], gi[2], Tf, whh, b_hh, h0, mode=mm_mode)
        return dict(m=m, graphs=graphs, loc_propose_pos=loc_propose_pos, scale_propose_pos=scale_propose_pos,
                    loc_propose_head=loc_propose_head, conc_propose_head=conc_propose_head, A=A,
                    edges=(ei_pl2m_1, ei_t2m, ei_pl2m, ei_a2m, ei_m2m))

    def _refine_phase(self, state: dict, temporal_m: Optional[torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Mode fusion hook + refinement stack + heads (reference :258-294)."""
        K, Tf, OD = self.num_modes, self.num_future_steps, self.output_dim
        A, graphs = state['A'], state['graphs']
        loc_propose_pos, scale_propose_pos = state['loc_propose_pos'], state['scale_propose_pos']
        loc_propose_head, conc_propose_head = state['loc_propose_head'], state['conc_propose_head']
        ei_pl2m_1, ei_t2m, ei_pl2m, ei_a2m, ei_m2m = state['edges']
        m = self._fuse_modes(state['m'], temporal_m)
        m = self._cross_stack('refine', m, *graphs)

        loc_ref, scale_ref, pi = MLPLayer.fused([self.to_loc_refine_pos, self.to_scale_refine_pos, self.to_pi], m,
                                                self._head_cache('refine'))
        if self.output_head:   # reference :271-274
            hloc, hconc = MLPLayer.fused([self.to_loc_refine_head, self.to_conc_refine_head], m, self._head_cache('refine_head'))
            loc_refine_head = torch.tanh(hloc.view(-1, K, Tf, 1)) * math.pi + loc_propose_head.detach()
            conc_refine_head = 1.0 / (F.elu(hconc.view(-1, K, Tf, 1)) + 1.0 + 0.02)
        loc_refine_pos = loc_ref.view(-1, K, Tf, OD) + loc_propose_pos.detach()
        scale_refine_pos = F.elu(scale_ref.view(-1, K, Tf, OD), alpha=1.0) + 1.0 + 0.1
        pi = pi.squeeze(-1)
        return {
            'loc_propose_pos': loc_propose_pos,
            'scale_propose_pos': scale_propose_pos,
            'loc_propose_head': loc_propose_head,
            'conc_propose_head': conc_propose_head,
            'loc_refine_pos': loc_refine_pos,
            'scale_refine_pos': scale_refine_pos,
            'loc_refine_head': loc_refine_head if self.output_head else loc_refine_pos.new_zeros((A, K, Tf, 1)),
            'conc_refine_head': conc_refine_head if self.output_head else scale_refine_pos.new_zeros((A, K, Tf, 1)),
            'pi': pi,
            self._mode_key: m,
            'edge_index_pl2m_clone': ei_pl2m_1,
            'edge_index_t2m': ei_t2m, 'edge_index_pl2m': ei_pl2m, 'edge_index_a2m': ei_a2m, 'edge_index_m2m': ei_m2m,
        }

    def forward(self, data, scene_enc: Mapping[str, torch.Tensor], time_shift: int,
                return_edge_index: bool = True) -> Dict[str, torch.Tensor]:
        """``return_edge_index=False`` (not in the reference): leave the K-fold replicated ``edge_index_*`` lists out of
        the result (None) — nothing downstream of the predictors reads them."""
        return self._decode(data, scene_enc, time_shift, None, return_edge_index)
