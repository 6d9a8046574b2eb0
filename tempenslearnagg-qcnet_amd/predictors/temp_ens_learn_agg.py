"""TempEnsLearnAgg — inference counterpart of predictors/temp_ens_learn_agg.py:22-282.

Keeps the sub-module names ``base_model`` (a frozen QCNetSliding, reference :26) and
``temporal_aggregate_layer`` (a TemporalAggregate initialised from the base decoder's weights, :61-81),
``forward(data, time_shift)`` (:123-126) and the frame loop of ``validation_step`` (:196-221) as ``predict``.
Training (losses, optimiser, Lightning hooks) and checkpoint download are outside the hot path: the model is
built from explicit hyper-parameters instead of ``load_from_checkpoint``.
"""
import os
import threading
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..metrics import MotionMetrics
from ..modules import TemporalAggregate
from ..utils.prep import inputs_ready, record_streams, register_stream, run_prep

_AGG_STREAMS: Dict[tuple, 'torch.cuda.Stream'] = {}     # per (host thread, device), like the preparation stream
_AGG_LOCK = threading.Lock()


def _aggregate_stream(device: torch.device) -> 'torch.cuda.Stream':
    idx = (threading.get_ident(), device.index if device.index is not None else torch.cuda.current_device())
    with _AGG_LOCK:
        if idx not in _AGG_STREAMS:
            _AGG_STREAMS[idx] = torch.cuda.Stream(device=device)
            register_stream(device, _AGG_STREAMS[idx])
        return _AGG_STREAMS[idx]
from .qcnet_sliding import QCNetSliding


class TempEnsLearnAgg(nn.Module):

    num_frames = 10  # time_shift = 1..10, reference :197

    def __init__(self, base_model: Optional[QCNetSliding] = None, share_map_encoding: bool = True, **hparams) -> None:
        super().__init__()
        self.base_model = base_model if base_model is not None else QCNetSliding(**hparams)
        hp = self.base_model.hparams
        for key, value in hp.items():
            setattr(self, key, value)
        self.temporal_aggregate_layer = TemporalAggregate(
            dataset=hp['dataset'], input_dim=hp['input_dim'], hidden_dim=hp['hidden_dim'], output_dim=hp['output_dim'],
            output_head=hp['output_head'], num_historical_steps=hp['num_historical_steps'],
            num_future_steps=hp['num_future_steps'], num_modes=hp['num_modes'],
            num_recurrent_steps=hp['num_recurrent_steps'], num_t2m_steps=hp['num_t2m_steps'],
            pl2m_radius=hp['pl2m_radius'], a2m_radius=hp['a2m_radius'], num_freq_bands=hp['num_freq_bands'],
            num_layers=hp['num_dec_layers'], num_heads=hp['num_heads'], head_dim=hp['head_dim'], dropout=hp['dropout'])
        self.temporal_aggregate_layer.load_state_dict(self.base_model.decoder.state_dict())
        # The map encoder does not depend on time_shift; the reference recomputes it for every frame.
        self.share_map_encoding = share_map_encoding
        # Consecutive frames share 49 of their 50 history steps: agent features, spatial neighbourhoods and their
        # positional embeddings depend on absolute time only (modules/qcnet_agent_encoder.py precompute).
        self.share_frame_work = share_map_encoding
        # The ten frames are independent given the shared work, so they run as ONE ten-fold batch: every layer is
        # one launch over all frames instead of ten small ones.
        self.batch_frames = share_map_encoding
        # Consecutive frames also share almost all of the agent encoder's LAYER outputs: its temporal attention is
        # causal with a reach of time_span steps, so a frame differs from one long run over all absolute steps only
        # in its first num_layers * time_span steps, of which the decoder reads at most a few
        # (modules/qcnet_agent_encoder.py forward_frames_streaming).
        self.stream_encoder = share_map_encoding
        # TemporalAggregate's proposal half beside the ten-frame decoder pass (QCN_OVERLAP_AGGREGATE=0: behind it)
        self.overlap_aggregate = os.environ.get('QCN_OVERLAP_AGGREGATE', '1') != '0'
        # Brier / minADE / minADE_best / minAHE / minFDE / minFHE / MR of the reference (:96-102), fed by one launch
        self.metrics = MotionMetrics(max_guesses=6)

    def get_ground_truth(self, data, time_shift: int = 10) -> torch.Tensor:
        """Regression targets in the agent frame of step T - 1 + time_shift (reference :104-121), one kernel."""
        T = self.num_historical_steps
        data['agent']['target'] = ops.to_agent_frame(data['agent']['position'].contiguous(), data['agent']['heading'].contiguous(),
                                                     T - 1 + time_shift, T + time_shift)
        return data['agent']['target']

    @torch.no_grad()
    def validation_step(self, data, batch_idx: int = 0) -> Dict[str, torch.Tensor]:
        """The evaluation half of the reference's validation_step (:188-282, losses excluded): prediction, ground truth,
        and the update of the seven metric states — eval-mask selection, motion-vector headings and all metric addends in
        one ``qcn_motion_metrics_f32`` launch.  Returns the predictions with the per-agent metric records."""
        pred = self.predict(data)
        reg_mask = data['agent']['predict_mask'][:, self.num_historical_steps + 10:]
        target = self.get_ground_truth(data, time_shift=10)
        eval_mask = data['agent']['category'] == 3
        pred['metric_records'] = self.metrics.update(pred['traj_refine'], pred['pi_softmax'], target, reg_mask, eval_mask,
                                                     output_dim=self.output_dim, output_head=bool(self.output_head))
        return pred

    def forward(self, data, time_shift: int, map_enc=None, agent_cache=None
                ) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        scene_enc = self.base_model.encoder(data, time_shift, map_enc=map_enc, agent_cache=agent_cache)
        pred = self.base_model.decoder(data, scene_enc, time_shift)
        return pred, scene_enc

    def _finish(self, pred: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """reference :204-220: [loc, (heading loc), scale, (heading concentration)] of the refined and proposed trajectories,
        truncated to the first 50 future steps."""
        od = self.output_dim
        if self.output_head:
            traj_propose = torch.cat([pred['loc_propose_pos'][..., :od], pred['loc_propose_head'],
                                      pred['scale_propose_pos'][..., :od], pred['conc_propose_head']], dim=-1)
            traj_refine = torch.cat([pred['loc_refine_pos'][..., :od], pred['loc_refine_head'],
                                     pred['scale_refine_pos'][..., :od], pred['conc_refine_head']], dim=-1)
        else:
            traj_propose = torch.cat([pred['loc_propose_pos'][..., :od], pred['scale_propose_pos'][..., :od]], dim=-1)
            traj_refine = torch.cat([pred['loc_refine_pos'][..., :od], pred['scale_refine_pos'][..., :od]], dim=-1)
        pred['traj_propose'] = traj_propose[..., :50, :]
        pred['traj_refine'] = traj_refine[..., :50, :]
        pred['pi_softmax'] = F.softmax(pred['pi'], dim=-1)
        return pred

    @torch.no_grad()
    def _predict_frames_batched(self, data) -> Dict[str, torch.Tensor]:
        from ..data import frame_batch
        T, n_f = self.num_historical_steps, self.num_frames
        enc, dec = self.base_model.encoder, self.base_model.decoder
        A = int(data['agent']['num_nodes'])
        dev = data['agent']['position'].device
        inputs_ready(dev, data, self)           # the preparation stream (utils/prep.py) may read `data` from here on
        map_enc = enc.map_encoder(data)
        cache = enc.agent_encoder.precompute(data, 1, n_f)
        # the frame batch replicates the map per frame; the decoder folds the polygon sources onto this one copy
        x_pl_frames = map_enc['x_pl']
        # (only what QCNetDecoder reads of its `data`: no map points, no velocities, no batch vectors)
        frames = run_prep(dev, lambda: frame_batch(data, 1, n_f, T + 2, fields=dec.DATA_FIELDS))
        if self.stream_encoder:
            Mt = min(dec.num_t2m_steps, T)
            table, index = enc.agent_encoder.stream_frames(data, map_enc, cache, T - Mt)           # [N, C], [F, A, Mt]
            scene = {'x_t_table': table, 'x_t_index': index.view(n_f * A, Mt), 'x_pl': x_pl_frames}
            scene_last = {'x_t_table': table, 'x_t_index': index[n_f - 1], 'x_pl': map_enc['x_pl']}
        else:
            x_a = enc.agent_encoder.forward_frames(data, map_enc, cache)                           # [F * A, T, C]
            scene = {'x_a': x_a, 'x_pl': x_pl_frames}
            scene_last = {'x_a': x_a[(n_f - 1) * A:], 'x_pl': map_enc['x_pl']}
        # The aggregation layer's proposal half (its own weights, the last frame's scene) does not depend on the ten frames'
        # mode queries: it runs on a stream of its own BESIDE the ten-frame decoder pass — one tenth of its rows per launch,
        # i.e. ~100 launches that fill a fraction of the GPU each and would otherwise run one after the other behind it.
        agg = self.temporal_aggregate_layer
        state = None
        if self.overlap_aggregate and dev.type == 'cuda':
            main = torch.cuda.current_stream(dev)
            side = _aggregate_stream(dev)
            side.wait_stream(main)                  # the encoder's outputs
            with torch.cuda.stream(side):
                state = agg.propose(data, scene_last, n_f, return_edge_index=False)
                done = side.record_event()
        pred = dec(frames, scene, 1, return_edge_index=False)
        temporal_m = pred['m'].view(n_f, A, self.num_modes, self.hidden_dim).transpose(0, 1)
        if state is None:
            state = agg.propose(data, scene_last, n_f, return_edge_index=False)
        else:
            record_streams(state, main)             # produced on the side stream, consumed (and freed) on the main stream
            record_streams(scene_last, side)
            main.wait_event(done)
        return self._finish(agg.refine(state, temporal_m))

    @torch.no_grad()
    def predict(self, data) -> Dict[str, torch.Tensor]:
        """Ten sliding QCNet frames + learned aggregation (reference :196-221, :261)."""
        if self.batch_frames and self.share_map_encoding and self.share_frame_work:
            return self._predict_frames_batched(data)
        inputs_ready(data['agent']['position'].device, data, self)
        map_enc = self.base_model.encoder.map_encoder(data) if self.share_map_encoding else None
        agent_cache = None
        if self.share_frame_work:
            agent_cache = self.base_model.encoder.agent_encoder.precompute(data, 1, self.num_frames)
        temporal_m = []
        scene_enc = None
        for time_shift in range(1, self.num_frames + 1):
            pred, scene_enc = self(data, time_shift, map_enc=map_enc, agent_cache=agent_cache)
            temporal_m.append(pred['m'])
        temporal_m = torch.stack(temporal_m, dim=1)
        return self._finish(self.temporal_aggregate_layer(temporal_m, data, scene_enc, self.num_frames))
