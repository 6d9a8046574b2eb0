"""QCNetSliding — inference counterpart of predictors/qcnet_sliding.py:56-226.

The reference class is a LightningModule (training loop, losses, metrics, optimiser): those are outside the hot
path.  This plain ``nn.Module`` keeps the constructor hyper-parameters (:58-87), the ``encoder`` / ``decoder``
sub-module names (so a reference state_dict loads unmodified), ``forward(data, time_shift)`` (:174-177) and
``postprocess`` (:179-197).
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..modules import QCNetDecoder, QCNetEncoder


class QCNetSliding(nn.Module):

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
                 num_freq_bands: int,
                 num_map_layers: int,
                 num_agent_layers: int,
                 num_dec_layers: int,
                 num_heads: int,
                 head_dim: int,
                 dropout: float,
                 pl2pl_radius: float,
                 time_span: Optional[int],
                 pl2a_radius: float,
                 a2a_radius: float,
                 num_t2m_steps: Optional[int],
                 pl2m_radius: float,
                 a2m_radius: float,
                 **kwargs) -> None:
        super().__init__()
        self.hparams = dict(
            dataset=dataset, input_dim=input_dim, hidden_dim=hidden_dim, output_dim=output_dim,
            output_head=output_head, num_historical_steps=num_historical_steps, num_future_steps=num_future_steps,
            num_modes=num_modes, num_recurrent_steps=num_recurrent_steps, num_freq_bands=num_freq_bands,
            num_map_layers=num_map_layers, num_agent_layers=num_agent_layers, num_dec_layers=num_dec_layers,
            num_heads=num_heads, head_dim=head_dim, dropout=dropout, pl2pl_radius=pl2pl_radius, time_span=time_span,
            pl2a_radius=pl2a_radius, a2a_radius=a2a_radius, num_t2m_steps=num_t2m_steps, pl2m_radius=pl2m_radius,
            a2m_radius=a2m_radius)
        for key, value in self.hparams.items():
            setattr(self, key, value)
        self.encoder = QCNetEncoder(
            dataset=dataset, input_dim=input_dim, hidden_dim=hidden_dim, num_historical_steps=num_historical_steps,
            pl2pl_radius=pl2pl_radius, time_span=time_span, pl2a_radius=pl2a_radius, a2a_radius=a2a_radius,
            num_freq_bands=num_freq_bands, num_map_layers=num_map_layers, num_agent_layers=num_agent_layers,
            num_heads=num_heads, head_dim=head_dim, dropout=dropout)
        self.decoder = QCNetDecoder(
            dataset=dataset, input_dim=input_dim, hidden_dim=hidden_dim, output_dim=output_dim,
            output_head=output_head, num_historical_steps=num_historical_steps, num_future_steps=num_future_steps,
            num_modes=num_modes, num_recurrent_steps=num_recurrent_steps, num_t2m_steps=num_t2m_steps,
            pl2m_radius=pl2m_radius, a2m_radius=a2m_radius, num_freq_bands=num_freq_bands, num_layers=num_dec_layers,
            num_heads=num_heads, head_dim=head_dim, dropout=dropout)

    def forward(self, data, time_shift: int, map_enc: Optional[Dict[str, torch.Tensor]] = None
                ) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        scene_enc = self.encoder(data, time_shift, map_enc=map_enc)
        pred = self.decoder(data, scene_enc, time_shift)
        return pred, scene_enc

    def postprocess(self, data, traj_refine: torch.Tensor, pi: torch.Tensor, time_shift: int):
        """Agent frame → world frame for the focal tracks, sliced to the common horizon
        (predictors/qcnet_sliding.py:179-197)."""
        T = self.num_historical_steps
        rows = (data['agent']['category'] == 3).nonzero(as_tuple=True)[0]
        traj = ops.to_world_frame(traj_refine, data['agent']['position'], data['agent']['heading'], T - 1 + time_shift, rows)
        return traj[:, :, 10 - time_shift: 10 - time_shift + 50], F.softmax(pi[rows], dim=-1)

    @torch.no_grad()
    def validation_step(self, data, batch_idx: int = 0) -> Dict[str, torch.Tensor]:
        """The sliding-window NMS baseline of the reference (:276-317): the frames time_shift = 10 .. 1 are predicted,
        moved to the world frame, and for every window size 1..10 the 6 * w candidates go through NMS and are scored
        against the future from step 60 on.  The reference does this per agent and window in numpy on the host
        (utils/nms.py); here every (agent, window) is one wave of ``qcn_nms_select_f32``.
        Returns ``selected`` [n_eval, 10, 6] (candidate indices), ``score`` [n_eval, 10, 3] (min FDE, min ADE, miss),
        ``single_score`` (the latest frame alone, :295-296) and ``miss_count_by_nms`` [10] (:316-317)."""
        trajs, clsfs = [], []
        for time_shift in range(10, 0, -1):
            pred, _ = self(data, time_shift)
            traj_refine = pred['loc_refine_pos'][..., :2]
            traj, pi = self.postprocess(data, traj_refine, pred['pi'], time_shift)
            trajs.append(traj)
            clsfs.append(pi)
        trajs = torch.stack(trajs, dim=1)                                                     # [n_eval, 10, K, 50, 2]
        clsfs = torch.stack(clsfs, dim=1)
        n_eval, n_w, k = trajs.shape[:3]
        eval_mask = data['agent']['category'] == 3
        gt = data['agent']['position'][eval_mask, 60:, :2].contiguous()
        selected, score = ops.nms_select(trajs.reshape(n_eval, n_w * k, trajs.size(3), 2), clsfs.reshape(n_eval, n_w * k), gt,
                                         cand_per_window=k, n_select=6)
        # the first frame alone, unfiltered (:295-296): all its K modes
        disp = torch.norm(trajs[:, 0] - gt.unsqueeze(1), p=2, dim=-1)
        single_fde = disp[..., -1].min(dim=-1).values
        single_miss = single_fde > 2.0
        miss_count = ((~single_miss).unsqueeze(1) & (score[..., 2] > 0)).sum(dim=0)
        return {'selected': selected, 'score': score, 'trajs': trajs, 'clsfs': clsfs,
                'single_score': torch.stack([single_fde, disp.mean(dim=-1).min(dim=-1).values, single_miss.float()], dim=-1),
                'miss_count_by_nms': miss_count}
