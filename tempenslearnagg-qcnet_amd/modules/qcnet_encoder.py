"""QCNetEncoder — drop-in for modules/qcnet_encoder.py:24-72."""
from typing import Dict, Optional

import torch
import torch.nn as nn

from .qcnet_agent_encoder import QCNetAgentEncoder
from .qcnet_map_encoder import QCNetMapEncoder


class QCNetEncoder(nn.Module):

    def __init__(self,
                 dataset: str,
                 input_dim: int,
                 hidden_dim: int,
                 num_historical_steps: int,
                 pl2pl_radius: float,
                 time_span: Optional[int],
                 pl2a_radius: float,
                 a2a_radius: float,
                 num_freq_bands: int,
                 num_map_layers: int,
                 num_agent_layers: int,
                 num_heads: int,
                 head_dim: int,
                 dropout: float) -> None:
        super().__init__()
        shared = dict(dataset=dataset, input_dim=input_dim, hidden_dim=hidden_dim,
                      num_historical_steps=num_historical_steps, num_freq_bands=num_freq_bands, num_heads=num_heads,
                      head_dim=head_dim, dropout=dropout)
        self.map_encoder = QCNetMapEncoder(pl2pl_radius=pl2pl_radius, num_layers=num_map_layers, **shared)
        self.agent_encoder = QCNetAgentEncoder(time_span=time_span, pl2a_radius=pl2a_radius, a2a_radius=a2a_radius,
                                               num_layers=num_agent_layers, **shared)

    def forward(self, data, time_shift: int, map_enc: Optional[Dict[str, torch.Tensor]] = None,
                agent_cache=None) -> Dict[str, torch.Tensor]:
        """``map_enc``: a map encoding computed earlier for the same scene.  The map encoder takes no
        ``time_shift`` (modules/qcnet_map_encoder.py:95), so the sliding loop may encode the map once.
        ``agent_cache``: ``agent_encoder.precompute(...)`` of the same scene (shift-invariant agent-side work)."""
        if map_enc is None:
            map_enc = self.map_encoder(data)
        agent_enc = self.agent_encoder(data, map_enc, time_shift, cache=agent_cache)
        return {**map_enc, **agent_enc}
