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
        eval_mask = data['agent']['category'] == 3
        origin = data['agent']['position'][eval_mask, T - 1 + time_shift]
        theta = data['agent']['heading'][eval_mask, T - 1 + time_shift]
        cos, sin = theta.cos(), theta.sin()
        rot = torch.stack([torch.stack([cos, sin], dim=-1), torch.stack([-sin, cos], dim=-1)], dim=-2)
        traj = torch.matmul(traj_refine[eval_mask, :, :, :2], rot.unsqueeze(1)) + origin[:, :2].reshape(-1, 1, 1, 2)
        return traj[:, :, 10 - time_shift: 10 - time_shift + 50], F.softmax(pi[eval_mask], dim=-1)
