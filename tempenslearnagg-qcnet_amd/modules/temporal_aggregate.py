"""TemporalAggregate — drop-in for modules/temporal_aggregate.py:35-301.

The reference class is a copy of QCNetDecoder (identical ctor and state_dict, so the decoder's weights load
into it, predictors/temp_ens_learn_agg.py:81) with one extra step: after the GRU re-encoding the new mode
queries are summed with the mode queries of the ten earlier frames (reference :261-263), and the refined
queries are returned as ``'final_m'`` (:299).
"""
from typing import Dict, Mapping, Optional

import torch

from .qcnet_decoder import QCNetDecoder


class TemporalAggregate(QCNetDecoder):

    _mode_key = 'final_m'

    def _fuse_modes(self, m: torch.Tensor, temporal_m: Optional[torch.Tensor]) -> torch.Tensor:
        K, C = self.num_modes, self.hidden_dim
        stacked = torch.cat([m.reshape(-1, 1, K, C), temporal_m], dim=1)   # [A, 1 + frames, K, C]
        return stacked.sum(dim=1).reshape(-1, C)

    def forward(self, temporal_m: torch.Tensor, data, scene_enc: Mapping[str, torch.Tensor],
                time_shift: int, return_edge_index: bool = True) -> Dict[str, torch.Tensor]:
        return self._decode(data, scene_enc, time_shift, temporal_m, return_edge_index)

    def propose(self, data, scene_enc: Mapping[str, torch.Tensor], time_shift: int, return_edge_index: bool = True) -> dict:
        """The half of ``forward`` that does not depend on ``temporal_m`` (reference :141-259); ``refine`` finishes it.
        ``forward(temporal_m, ...) == refine(propose(...), temporal_m)``."""
        return self._propose_phase(data, scene_enc, time_shift, return_edge_index)

    def refine(self, state: dict, temporal_m: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self._refine_phase(state, temporal_m)

